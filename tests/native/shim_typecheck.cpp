// type-check only (g++ -fsyntax-only): the reference-side binding against the transcribed declarations
#include "ref_decls.hpp"
#include "isvins_estimator_shim.hpp"
void typecheck_uses(Estimator &e) {
    e.isv_handle = isvins::create_backend();
    Estimator_initFactorGraph_isv(e);
    Estimator_backendOptimization_isv(e);
}
