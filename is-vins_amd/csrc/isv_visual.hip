// isv_visual.hip -- k_lin_gram: ProjectionFactor::Evaluate (+ CauchyLoss corrector) FUSED with the direct part of the
// landmark elimination (the Gram products k_sweep_mfma forms), for the solver path.
//
// k_proj_linearize<0> wrote a 224-byte Jacobian strip per factor to HBM and k_sweep_mfma read it straight back
// (0.45 GB per iteration of a 1024-window batch: more than half of the step's HBM traffic).  Here the strips never
// leave the CU: the factors of a window are visited in (host, observer) frame-PAIR order (pg_* arrays, built at
// upload), one LANE per factor evaluates residual and Jacobians in registers (src/factor/projection_factor.cpp:24-122),
// 16 factors at a time are laid out in a wave-private LDS tile and fed to v_mfma_f64_16x16x4 exactly as
// k_sweep_mfma fed them from HBM:   G = X^T X,  X = [J_i | J_j | r]  (2 rows per factor, 13 columns).
// What still goes to HBM per factor: its cost (8 B), w = J_j^T J_l of the observing frame (48 B, the packed W layout
// of the rank-1 kernel) and a 64-byte record {J_l^T J_l, J_l^T r, J_i^T J_l} from which k_rank1_mfma's prologue forms
// the landmark scalars in the landmark's own factor order (fixed order: bitwise reproducible, no atomics).
//
// One workgroup (LGW wavefronts) per window; wavefront v takes the pair groups the upload schedule gave to the
// sweep wavefronts 2v and 2v + 1, laid out back to back in the factor stream (pg_rec / pg_pts / pg_wstart) so that it
// walks them in FULL 64-lane chunks.  Outputs: Tvis (pose blocks, gradient, Jacobi diagonal) as k_sweep_mfma.
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"
#include "isv_proj_factor.h"
#include "isv_lin_gram.h"

// lcap: landmarks per window the launch stages (inverse depths + host points: 4 doubles each)
size_t lin_gram_lds_bytes(int N /* real frames */, bool partials_in_lds, bool ex, int waves, int lcap) {
    const size_t NP = (size_t)N * (N - 1) / 2;
    return ((size_t)N * 12 + 12 + (size_t)waves * 16 * lg_xld(ex) + (partials_in_lds ? NP * 84 : 0) + (NP + 2) / 2 + 1 + (NP + 1) / 2 + 1 + 4 * (size_t)lcap) * sizeof(double);
}

template <bool EX, int LGW>
__global__ __launch_bounds__(64 * LGW, LGW > 4 ? 2 : (EX ? 2 : 3)) void k_lin_gram(DevBatch d) {
    extern __shared__ __align__(16) double lds[];
    lin_gram_body<EX, LGW>(d, lds);
}
template __global__ void k_lin_gram<false, LG_WAVES>(DevBatch);
template __global__ void k_lin_gram<true, LG_WAVES>(DevBatch);
template __global__ void k_lin_gram<false, LG_WAVES_SMALL>(DevBatch);
template __global__ void k_lin_gram<true, LG_WAVES_SMALL>(DevBatch);
