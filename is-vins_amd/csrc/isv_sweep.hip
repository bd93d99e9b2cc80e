// isv_sweep.hip -- batch-wide landmark elimination kernels (the SchurEliminator part of Ceres'
// DENSE_SCHUR, for the problem problemSolve() builds, reference src/estimator.cpp:1057-1092).
//
//   k_lm_prep   one lane per landmark: E_l = J_l^T J_l, g_l = J_l^T r, host-frame w, Jacobi scale
//               (iteration 0), dogleg diagonal, scaled gradient and the elimination weight
//               c_l = s_l^2 / (s_l^2 E_l + mu D_l^2)  (stored with g_l as one 16-byte record).
//   k_sweep     one WAVEFRONT per (window, frame a): block column a of the reprojection part of the
//               reduced matrix,  sum_l [ J_p^T J_p - c_l w w^T ],  kept in registers (lane = block row
//               (bo, r), 6 entries per lane) over all landmarks covering frame a, in landmark order
//               (owner-computes => bitwise reproducible), then written once (6x6 pose corners only:
//               19 KB per 11-frame window).  B*N independent wavefronts keep every SIMD busy, unlike a
//               per-window workgroup sweep that is bound by one CU's issue rate.
//   k_backsub   one lane per landmark: back-substitution + Cauchy-point terms from the w vectors.
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"

DEV int win_of_landmark(const DevBatch &d, int l) {
    int lo = 0, hi = d.B;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (d.lm_off[mid] <= l) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256) void k_lm_prep(DevBatch d) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= d.Ltot) return;
    const int w = win_of_landmark(d, l);
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int k = d.lm_k[l], f0 = d.lm_f0[l];
    double E = 0, gl = 0, wh[6] = {0, 0, 0, 0, 0, 0};
    for (int m = 0; m < k - 1; m++) {
        const double *s = d.strip + (size_t)(f0 + m) * ISV_PROJ_STRIP;
        const double2 rs = *reinterpret_cast<const double2 *>(s), jl = *reinterpret_cast<const double2 *>(s + 26);
        E += jl.x * jl.x + jl.y * jl.y; gl += jl.x * rs.x + jl.y * rs.y;
#pragma unroll
        for (int c = 0; c < 6; c++) wh[c] += s[2 + c] * jl.x + s[8 + c] * jl.y;
    }
    double sl;
    if (st.iteration == 0) { sl = 1.0 / (1.0 + sqrt(E)); d.scale_l[l] = sl; }
    else sl = d.scale_l[l];
    const double Es = sl * sl * E;
    const double Dl2 = fmin(fmax(Es, 1e-6), 1e32);
    const double Dl = sqrt(Dl2);
    d.lm_cg[l] = make_double2(sl * sl / (Es + st.mu * Dl2), gl);
    d.lmE[l] = E; d.lmG[l] = gl; d.diag_l[l] = Dl; d.grad_l[l] = sl * gl / Dl;
    double *wo = d.W + (size_t)(f0 + l) * 6;           // host observation slot
#pragma unroll
    for (int c = 0; c < 6; c++) wo[c] = wh[c];
}

// Tvis layout of one window: block column a at 36 * (a N - a (a-1) / 2), block (a+bo, a) = 36 doubles
// row-major 6x6; then hd[6N] (diag of the direct part), g[6N], bs[6N].
__host__ __device__ inline int tvis_col(int a, int N) { return 36 * (a * N - a * (a - 1) / 2); }

__global__ __launch_bounds__(64) void k_sweep(DevBatch d) {
    const int w = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, l0 = d.lm_off[w], l1 = d.lm_off[w + 1], fw0 = d.f_off[w];
    // lane = block row (bo, r) of column a, two steps cover bo <= 20.  Every lane touches at most ONE
    // factor per landmark, so all loads of a landmark are issued as one batch:
    //   acc : rows of block (a+bo, a);   hh : per-factor partials of the host block (a, a), summed over
    //   bo in fixed order at the end (lanes bo >= 1 hold the partial of factor bo-1).
    double acc[12], hh[12], hdp[2] = {0, 0}, ghp[2] = {0, 0}, hd = 0, gacc = 0, bacc = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { acc[i] = 0; hh[i] = 0; }
    const int bo0 = lane / 6, r0 = lane - 6 * bo0;                 // step 0
    const int bo1 = (lane + 64) / 6, r1 = (lane + 64) - 6 * bo1;   // step 1
    for (int base = l0; base < l1; base += 64) {
        const unsigned mm = (base + lane < l1) ? d.lm_meta[base + lane] : 0u;
        const int hh_ = mm & 255, kk = (mm >> 8) & 255;
        unsigned long long mask = __ballot(base + lane < l1 && a >= hh_ && a < hh_ + kk);
        while (mask) {
            const int bit = __builtin_ctzll(mask);
            mask &= mask - 1;
            const unsigned m0 = __builtin_amdgcn_readlane(mm, bit);
            const int l = base + bit, h = m0 & 255, k = (m0 >> 8) & 255;
            const int f0 = fw0 + (int)(m0 >> 16), pa = a - h, nb = k - pa;
            const double *wa = d.W + (size_t)(f0 + l + pa) * 6;
            const bool host = (pa == 0);
            const int cofs = host ? 2 : 14;              // column block: Ji (host column) or Jj
#pragma unroll
            for (int step = 0; step < 2; step++) {
                if (step == 0 || 64 < 6 * nb) {          // step 1 only for > 10-frame tracks (wave-uniform)
                    const int bo = step ? bo1 : bo0, r = step ? r1 : r0;
                    // branch-free body: every load is issued up front (clamped addresses), the lane's role
                    // enters through zeroed multipliers, so one memory round trip per landmark
                    const bool act = bo < nb;
                    const bool has_f = act && (host ? (bo > 0) : (bo == 0));
                    const int fidx = has_f ? (host ? bo - 1 : pa - 1) : 0;
                    const double *s = d.strip + (size_t)(f0 + fidx) * ISV_PROJ_STRIP;
                    const double2 cg = d.lm_cg[l];
                    const double2 wa01 = *reinterpret_cast<const double2 *>(wa), wa23 = *reinterpret_cast<const double2 *>(wa + 2),
                                  wa45 = *reinterpret_cast<const double2 *>(wa + 4);
                    const double wbr = wa[(act ? bo : 0) * 6 + r];
                    const double jj0 = s[14 + r], jj1 = s[20 + r], ji0 = s[2 + r], ji1 = s[8 + r];
                    const double2 c01 = *reinterpret_cast<const double2 *>(s + cofs), c23 = *reinterpret_cast<const double2 *>(s + cofs + 2),
                                  c45 = *reinterpret_cast<const double2 *>(s + cofs + 4), d01 = *reinterpret_cast<const double2 *>(s + cofs + 6),
                                  d23 = *reinterpret_cast<const double2 *>(s + cofs + 8), d45 = *reinterpret_cast<const double2 *>(s + cofs + 10);
                    const double2 rs = *reinterpret_cast<const double2 *>(s);
                    const double coef = act ? -cg.x * wbr : 0.0;
                    const double fj0 = has_f ? jj0 : 0.0, fj1 = has_f ? jj1 : 0.0;
                    const double hi0 = (has_f && host) ? ji0 : 0.0, hi1 = (has_f && host) ? ji1 : 0.0;
                    const double dj0 = host ? 0.0 : fj0, dj1 = host ? 0.0 : fj1;
                    double *ac = acc + 6 * step, *hp = hh + 6 * step;
                    ac[0] += coef * wa01.x + fj0 * c01.x + fj1 * d01.x; ac[1] += coef * wa01.y + fj0 * c01.y + fj1 * d01.y;
                    ac[2] += coef * wa23.x + fj0 * c23.x + fj1 * d23.x; ac[3] += coef * wa23.y + fj0 * c23.y + fj1 * d23.y;
                    ac[4] += coef * wa45.x + fj0 * c45.x + fj1 * d45.x; ac[5] += coef * wa45.y + fj0 * c45.y + fj1 * d45.y;
                    hp[0] += hi0 * c01.x + hi1 * d01.x; hp[1] += hi0 * c01.y + hi1 * d01.y;
                    hp[2] += hi0 * c23.x + hi1 * d23.x; hp[3] += hi0 * c23.y + hi1 * d23.y;
                    hp[4] += hi0 * c45.x + hi1 * d45.x; hp[5] += hi0 * c45.y + hi1 * d45.y;
                    hdp[step] += hi0 * hi0 + hi1 * hi1;
                    ghp[step] += hi0 * rs.x + hi1 * rs.y;
                    hd += dj0 * dj0 + dj1 * dj1;
                    gacc += dj0 * rs.x + dj1 * rs.y;
                    bacc += (act && bo == 0) ? coef * cg.y : 0.0;
                }
            }
        }
    }
    // fold the host-block partials (lanes bo >= 1) into the lanes bo == 0, in ascending bo
    __shared__ double red[2 * 64 * 8];
#pragma unroll
    for (int step = 0; step < 2; step++) {
        double *o = red + (step * 64 + lane) * 8;
#pragma unroll
        for (int c = 0; c < 6; c++) o[c] = hh[6 * step + c];
        o[6] = hdp[step]; o[7] = ghp[step];
    }
    __syncthreads();
    if (bo0 == 0) {
        for (int bo = 1; a + bo < N; bo++) {
            const int ll = 6 * bo + r0;
            const double *o = red + ll * 8;              // ll < 128: (step, lane) laid out contiguously
#pragma unroll
            for (int c = 0; c < 6; c++) acc[c] += o[c];
            hd += o[6]; gacc += o[7];
        }
    }
    double *out = d.Tvis + (size_t)w * d.tvis_sz;
    const int colbase = tvis_col(a, N), tail = 36 * (N * (N + 1) / 2);
#pragma unroll
    for (int step = 0; step < 2; step++) {
        const int bo = step ? bo1 : bo0, r = step ? r1 : r0;
        if (a + bo < N) {
            double *o = out + colbase + bo * 36 + r * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) o[c] = acc[6 * step + c];
            if (bo == 0) { out[tail + 6 * a + r] = hd; out[tail + 6 * N + 6 * a + r] = gacc; out[tail + 12 * N + 6 * a + r] = bacc; }
        }
    }
}

// back-substitution of the eliminated landmarks (schur_eliminator BackSubstitute) + the landmark
// terms of the Cauchy-point denominator, from the w vectors (48 B per observation).
__global__ __launch_bounds__(256) void k_backsub(DevBatch d) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= d.Ltot) return;
    const int w = win_of_landmark(d, l);
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.fresh || st.ls_fail) return;
    const int n = d.np, h = d.lm_host[l], k = d.lm_k[l], f0 = d.lm_f0[l];
    const double *zp = d.zp + (size_t)w * n, *up = d.up + (size_t)w * n;
    const double *wv = d.W + (size_t)(f0 + l) * 6;
    double wz = 0, wu = 0;        // w_l^T z_p, w_l^T u_p
    for (int o = 0; o < k; o++) {
        const double2 w01 = *reinterpret_cast<const double2 *>(wv + 6 * o), w23 = *reinterpret_cast<const double2 *>(wv + 6 * o + 2),
                      w45 = *reinterpret_cast<const double2 *>(wv + 6 * o + 4);
        const double *z = zp + 15 * (h + o), *u = up + 15 * (h + o);
        wz += w01.x * z[0] + w01.y * z[1] + w23.x * z[2] + w23.y * z[3] + w45.x * z[4] + w45.y * z[5];
        wu += w01.x * u[0] + w01.y * u[1] + w23.x * u[2] + w23.y * u[3] + w45.x * u[4] + w45.y * u[5];
    }
    const double sl = d.scale_l[l], E = d.lmE[l], gl = d.lmG[l], Dl = d.diag_l[l];
    const double Es = sl * sl * E, Dl2 = Dl * Dl;
    // scaled-space y_l = (g'_l - w'_l^T y_p) / (E'_l + mu D_l^2),  w'^T y_p = s_l w^T (Sc_p y_p) = s_l wz
    const double yl = (sl * gl - sl * wz) / (Es + st.mu * Dl2);
    d.gn_l[l] = -Dl * yl;
    const double ul = sl * sl * gl / Dl2, cl = sl * sl / (Es + st.mu * Dl2);
    d.lm_aterm[l] = cl * wu * wu + 2.0 * ul * wu + E * ul * ul;
}
