// isv_solver.hip -- (stub, replaced by the on-device trust-region solve)
#include "isv_kernels.h"
int isv_solver_alloc(DevBatch &, size_t, size_t, size_t, std::vector<void *> &, std::string &) { return ISV_OK; }
int isv_solver_enqueue(DevBatch &, hipStream_t, int64_t *, std::string &err) { err = "solver not built"; return ISV_ERR_UNSUPPORTED; }
int isv_solver_download(DevBatch &, hipStream_t, int, isv_summary_t *, isv_marg_result_t *, std::string &) { return ISV_OK; }
int isv_solver_debug_read(DevBatch &, hipStream_t, int, double *, int64_t, std::string &) { return ISV_ERR_INVALID_ARG; }
