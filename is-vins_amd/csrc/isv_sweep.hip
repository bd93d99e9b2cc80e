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
    double *wd = d.Wd + (size_t)l * d.wd_ld + 6 * d.lm_host[l];
#pragma unroll
    for (int c = 0; c < 6; c++) { wo[c] = wh[c]; wd[c] = wh[c]; }
}

// Tvis layout of one window: block column a at 36 * (a N - a (a-1) / 2), block (a+bo, a) = 36 doubles
// row-major 6x6; then hd[6N] (diag of the direct part), g[6N], bs[6N].
__host__ __device__ inline int tvis_col(int a, int N) { return 36 * (a * N - a * (a - 1) / 2); }

// DIRECT part of block column a: sum over the reprojection factors of J_p^T J_p (no landmark coupling;
// the rank-1 downdates - c_l w w^T come from k_rank1_mfma).  Lane = (group g = lane/6, row r = lane%6):
//   * a landmark hosted in frame a ("host" pair): group g >= 1 works on factor g-1: block (a+g, a) +=
//     J_j^T J_i, and the partial J_i^T J_i of the host block (a, a);
//   * landmarks that merely observe frame a: TEN of them per iteration, group g takes the g-th one and
//     adds its J_j^T J_j to its partial of block (a, a).
// The per-group partials of block (a, a) are folded in ascending group order at the end (fixed order =>
// bitwise reproducible).
__global__ __launch_bounds__(64) void k_sweep(DevBatch d) {
    const int w = blockIdx.x, a = blockIdx.y, lane = threadIdx.x;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, l0 = d.lm_off[w], l1 = d.lm_off[w + 1], fw0 = d.f_off[w];
    double acc[12], dgp[12], hdp[2] = {0, 0}, ghp[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < 12; i++) { acc[i] = 0; dgp[i] = 0; }
    const int g0 = lane / 6, r0 = lane - 6 * g0;
    const int g1 = (lane + 64) / 6, r1 = (lane + 64) - 6 * g1;

    auto factor_update = [&](const double *s, int rofs, int cofs, int r, bool on, double *blk, double *dg, double &hd, double &gh, bool host) {
        // rows: J[rofs + r], J[rofs + 6 + r]; column block at cofs (2 x 6); host: also the J_i^T J_i partial
        const double x0 = s[rofs + r], x1 = s[rofs + 6 + r];
        const double y0 = s[2 + r], y1 = s[8 + r];                 // J_i rows (host partial)
        const double2 c01 = *reinterpret_cast<const double2 *>(s + cofs), c23 = *reinterpret_cast<const double2 *>(s + cofs + 2),
                      c45 = *reinterpret_cast<const double2 *>(s + cofs + 4), d01 = *reinterpret_cast<const double2 *>(s + cofs + 6),
                      d23 = *reinterpret_cast<const double2 *>(s + cofs + 8), d45 = *reinterpret_cast<const double2 *>(s + cofs + 10);
        const double2 rs = *reinterpret_cast<const double2 *>(s);
        const double f0 = on ? x0 : 0.0, f1 = on ? x1 : 0.0;
        if (host) {
            const double h0 = on ? y0 : 0.0, h1 = on ? y1 : 0.0;
            blk[0] += f0 * c01.x + f1 * d01.x; blk[1] += f0 * c01.y + f1 * d01.y; blk[2] += f0 * c23.x + f1 * d23.x;
            blk[3] += f0 * c23.y + f1 * d23.y; blk[4] += f0 * c45.x + f1 * d45.x; blk[5] += f0 * c45.y + f1 * d45.y;
            dg[0] += h0 * c01.x + h1 * d01.x; dg[1] += h0 * c01.y + h1 * d01.y; dg[2] += h0 * c23.x + h1 * d23.x;
            dg[3] += h0 * c23.y + h1 * d23.y; dg[4] += h0 * c45.x + h1 * d45.x; dg[5] += h0 * c45.y + h1 * d45.y;
            hd += h0 * h0 + h1 * h1; gh += h0 * rs.x + h1 * rs.y;
        } else {
            dg[0] += f0 * c01.x + f1 * d01.x; dg[1] += f0 * c01.y + f1 * d01.y; dg[2] += f0 * c23.x + f1 * d23.x;
            dg[3] += f0 * c23.y + f1 * d23.y; dg[4] += f0 * c45.x + f1 * d45.x; dg[5] += f0 * c45.y + f1 * d45.y;
            hd += f0 * f0 + f1 * f1; gh += f0 * rs.x + f1 * rs.y;
        }
    };

    for (int base = l0; base < l1; base += 64) {
        const unsigned mm = (base + lane < l1) ? d.lm_meta[base + lane] : 0u;
        const int hh_ = mm & 255, kk = (mm >> 8) & 255;
        const bool cover = base + lane < l1 && a >= hh_ && a < hh_ + kk;
        unsigned long long maskH = __ballot(cover && hh_ == a), maskN = __ballot(cover && hh_ != a);
        while (maskH) {                                   // landmarks hosted in frame a, one per iteration
            const int bit = __builtin_ctzll(maskH);
            maskH &= maskH - 1;
            const unsigned m0 = __builtin_amdgcn_readlane(mm, bit);
            const int k = (m0 >> 8) & 255, f0 = fw0 + (int)(m0 >> 16);
            {
                const bool on = g0 >= 1 && g0 < k;
                const double *s = d.strip + (size_t)(f0 + (on ? g0 - 1 : 0)) * ISV_PROJ_STRIP;
                factor_update(s, 14, 2, r0, on, acc, dgp, hdp[0], ghp[0], true);
            }
            if (k > 10) {                                  // groups 10.. live in the second lane set
                const bool on = g1 >= 1 && g1 < k;
                const double *s = d.strip + (size_t)(f0 + (on ? g1 - 1 : 0)) * ISV_PROJ_STRIP;
                factor_update(s, 14, 2, r1, on, acc + 6, dgp + 6, hdp[1], ghp[1], true);
            }
        }
        while (maskN) {                                   // observers of frame a, ten per iteration
            int mybit = 0; bool on = false;
#pragma unroll
            for (int sl = 0; sl < 10; sl++) {
                if (maskN) {
                    const int bit = __builtin_ctzll(maskN);
                    maskN &= maskN - 1;
                    if (g0 == sl) { mybit = bit; on = true; }
                }
            }
            const unsigned m0 = __shfl(mm, mybit);
            const int h = m0 & 255, f0 = fw0 + (int)(m0 >> 16);
            const double *s = d.strip + (size_t)(on ? f0 + (a - h) - 1 : fw0) * ISV_PROJ_STRIP;
            factor_update(s, 14, 14, r0, on, acc, dgp, hdp[0], ghp[0], false);
        }
    }
    // fold the per-group partials of block (a, a), its diagonal and the gradient into group 0
    __shared__ double red[128 * 8];
#pragma unroll
    for (int step = 0; step < 2; step++) {
        double *o = red + (step * 64 + lane) * 8;
#pragma unroll
        for (int c = 0; c < 6; c++) o[c] = dgp[6 * step + c];
        o[6] = hdp[step]; o[7] = ghp[step];
    }
    __syncthreads();
    double hd = 0, gacc = 0;
    if (g0 == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) acc[c] = 0;             // group 0 has no off-diagonal block of its own
        for (int g = 0; g < 21; g++) {                       // 126 lane slots = 21 groups
            const double *o = red + (6 * g + r0) * 8;
#pragma unroll
            for (int c = 0; c < 6; c++) acc[c] += o[c];
            hd += o[6]; gacc += o[7];
        }
    }
    double *out = d.Tvis + (size_t)w * d.tvis_sz;
    const int colbase = tvis_col(a, N), tail = 36 * (N * (N + 1) / 2);
#pragma unroll
    for (int step = 0; step < 2; step++) {
        const int bo = step ? g1 : g0, r = step ? r1 : r0;
        if (a + bo < N) {
            double *o = out + colbase + bo * 36 + r * 6;
#pragma unroll
            for (int c = 0; c < 6; c++) o[c] = acc[6 * step + c];
            if (bo == 0) { out[tail + 6 * a + r] = hd; out[tail + 6 * N + 6 * a + r] = gacc; }
        }
    }
}

// DIRECT part as FP64 MFMA Gram products.  The factors of a window are sorted by (host h, observer j)
// frame pair at upload (pg_perm / pg_off).  For one pair the stacked rows X = [J_i | J_j | r] (2 rows
// per factor, 13 columns) give, in ONE accumulator tile G = X^T X (v_mfma_f64_16x16x4, two factors per
// instruction, A and B operands are the same register):
//     G[0:6, 0:6] = sum J_i^T J_i  (part of block (h,h))     G[6:12, 0:6]  = sum J_j^T J_i = block (j,h)
//     G[6:12,6:12] = sum J_j^T J_j (part of block (j,j))     G[0:6,12], G[6:12,12] = J_i^T r, J_j^T r
// The pair groups are spread over the 16 wavefronts of the workgroup by a longest-first schedule built
// at upload (pg_sched); every group leaves its five pieces in LDS / Tvis, then the (a,a) blocks, the
// Jacobi diagonal and the gradient are folded in a fixed order (bitwise reproducible, no atomics).
// Every strip is read exactly once (the scalar k_sweep read it twice and was issue / latency bound).
typedef double double4s __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64 * ISV_SWEEP_WAVES) void k_sweep_mfma(DevBatch d) {
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, NP = N * (N - 1) / 2;
    double *Pjj = lds;                     // [NP][36]  sum J_j^T J_j of pair p
    double *Phh = Pjj + NP * 36;           // [NP][36]  sum J_i^T J_i of pair p
    double *Pgj = Phh + NP * 36;           // [NP][6]   sum J_j^T r
    double *Pgh = Pgj + NP * 6;            // [NP][6]   sum J_i^T r
    int *offL = (int *)(Pgh + NP * 6);     // [NP + 1] group starts, staged once
    const int *perm = d.pg_perm + d.f_off[w];
    const int *sched = d.pg_sched + (size_t)w * NP, *soff = d.pg_sched_off + (size_t)w * (ISV_SWEEP_WAVES + 1);
    const double *strip = d.strip + (size_t)d.f_off[w] * ISV_PROJ_STRIP;
    double *out = d.Tvis + (size_t)w * d.tvis_sz;
    const int i = lane & 15, kq = lane >> 4, row2 = kq & 1, fsel = kq >> 1;
    const int eoff = i < 6 ? 2 + row2 * 6 + i : (i < 12 ? 14 + row2 * 6 + (i - 6) : row2);   // strip element of operand column i
    const bool colok = i < 13;
#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64();
#define SWSTAMP(k) do { if (t == 0) { unsigned long long now_ = wall_clock64(); d.dbg[(size_t)w * 64 + (k)] += (double)(now_ - t_last); t_last = now_; } } while (0)
#else
#define SWSTAMP(k) do {} while (0)
#endif
    const int q0 = soff[wv], q1 = soff[wv + 1];
    const int myrec = (q0 + lane < q1) ? sched[q0 + lane] : 0;         // this wavefront's groups (<= 64)
    for (int e = t; e <= NP; e += blockDim.x) offL[e] = d.pg_off[(size_t)w * (NP + 1) + e];
    __syncthreads();
    SWSTAMP(40);
    // first chunk of the first group; the next group's chunk is prefetched while the current one is consumed
    int nxt = 0;
    if (q0 < q1) { const int p = __shfl(myrec, 0) >> 16; const int b0 = offL[p], c0 = offL[p + 1] - b0; nxt = lane < c0 ? perm[b0 + lane] : 0; }
    for (int q = q0; q < q1; q++) {
        const int rec = __shfl(myrec, q - q0), h = rec & 255, j = (rec >> 8) & 255, p = rec >> 16;
        const int b0 = offL[p], b1 = offL[p + 1];
        int myf = nxt;
        if (q + 1 < q1) { const int p2 = __shfl(myrec, q + 1 - q0) >> 16; const int b2 = offL[p2], c2 = offL[p2 + 1] - b2; nxt = lane < c2 ? perm[b2 + lane] : 0; }
        double4s acc = {0, 0, 0, 0};
        for (int base = b0; base < b1; base += 64) {
            const int cnt = (b1 - base) < 64 ? (b1 - base) : 64;
            if (base > b0) myf = lane < cnt ? perm[base + lane] : 0;
            for (int s2 = 0; s2 < cnt; s2 += 16) {                     // 16 factors: 8 loads in flight, then 8 MFMAs
                double v[8];
#pragma unroll
                for (int u2 = 0; u2 < 8; u2++) {
                    const int src = s2 + 2 * u2 + fsel;
                    const int f = __shfl(myf, src & 63);
                    v[u2] = (colok && src < cnt) ? strip[(size_t)f * ISV_PROJ_STRIP + eoff] : 0.0;
                }
#pragma unroll
                for (int u2 = 0; u2 < 8; u2++)
                    if (s2 + 2 * u2 < cnt) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u2], v[u2], acc, 0, 0, 0);
            }
        }
        // C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
        for (int reg = 0; reg <= 2; reg++) {
            const int row = kq + 4 * reg;
            if (row < 6) {
                if (i < 6) Phh[p * 36 + row * 6 + i] = acc[reg];
                else if (i == 12) Pgh[p * 6 + row] = acc[reg];
            } else if (row < 12) {
                const int rr = row - 6;
                if (i < 6) out[tvis_col(h, N) + (j - h) * 36 + rr * 6 + i] = acc[reg];        // block (j, h)
                else if (i < 12) Pjj[p * 36 + rr * 6 + (i - 6)] = acc[reg];
                else if (i == 12) Pgj[p * 6 + rr] = acc[reg];
            }
        }
    }
    SWSTAMP(41);
    __syncthreads();
    SWSTAMP(42);
    const int tail = 36 * (N * (N + 1) / 2);
    auto pidx = [N](int hh, int jj) { return hh * N - hh * (hh + 1) / 2 + (jj - hh - 1); };
    if (t < N * 36) {                      // diagonal blocks and the Jacobi-scaling diagonal
        const int a = t / 36, rc = t - 36 * a, r = rc / 6, c = rc - 6 * r;
        double s = 0.0;
        for (int j2 = a + 1; j2 < N; j2++) s += Phh[pidx(a, j2) * 36 + rc];
        for (int h2 = 0; h2 < a; h2++) s += Pjj[pidx(h2, a) * 36 + rc];
        out[tvis_col(a, N) + rc] = s;
        if (r == c) out[tail + 6 * a + r] = s;
    } else if (t < N * 42) {               // gradient
        const int q = t - N * 36, a = q / 6, r = q - 6 * a;
        double s = 0.0;
        for (int j2 = a + 1; j2 < N; j2++) s += Pgh[pidx(a, j2) * 6 + r];
        for (int h2 = 0; h2 < a; h2++) s += Pgj[pidx(h2, a) * 6 + r];
        out[tail + 6 * N + 6 * a + r] = s;
    }
    SWSTAMP(43);
}

// Rank-1 landmark downdates as FP64 MFMA panels:  Tvis -= P^T diag(c) P  over the window's landmarks,
// where row l of the panel P is the landmark's w vector dense over the 6N pose columns (zero where a frame
// does not see it) with g_l appended in column 6N.  One wavefront per 16x16 output tile (lower triangle,
// zero padded to 16s), v_mfma_f64_16x16x4: A = c_l * P[l][16I + i] for 4 landmarks, B = P[l][16J + j].
// The dense panels multiply structural zeros, but one MFMA replaces ~1000 scalar lane-FMAs with their
// address arithmetic (the scalar sweep was issue bound).  Row 6N of the product is sum c_l g_l w_l = -bs,
// the reduced right-hand side.  The packed w vectors (HBM) are expanded to panel rows in LDS, 64 landmarks
// per pass; the loads of the next pass are in flight while the current one is multiplied.
typedef double double4v __attribute__((ext_vector_type(4)));
#define R1_CHUNK 64                       // landmarks staged per pass (64 x wd_ld doubles of LDS)
// NT = panel width / 16 (compile time: cheap index arithmetic, right-sized prefetch registers)
template <int NT>
__global__ __launch_bounds__(64 * NT * (NT + 1) / 2) void k_rank1_mfma(DevBatch d) {
    constexpr int R1_PF = (32 + NT) / (NT + 1);            // panel elements per thread and pass: 64 * ld / threads = 32 / (NT + 1)
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    constexpr int ld = 16 * NT, nthr = 64 * NT * (NT + 1) / 2;
    const int N = d.N, n6 = 6 * N;
    const int l0 = d.lm_off[w], l1 = d.lm_off[w + 1], Lw = l1 - l0;
    constexpr int lds_ld = ld + 4;             // padded rows: the 4 k-rows of an operand hit distinct banks
    double *sW = lds;                      // [R1_CHUNK][ld + 4]
    double2 *sCG = (double2 *)(lds + R1_CHUNK * lds_ld);       // [max_lm] {c_l, g_l}
    unsigned *sM = (unsigned *)(sCG + d.max_lm);               // [max_lm] landmark metadata
    const int fw0 = d.f_off[w];
    double *out = d.Tvis + (size_t)w * d.tvis_sz;
    const int tail = 36 * (N * (N + 1) / 2);
    // wavefront wv owns output tile (I, J), I >= J
    int I = 0;
    while ((I + 1) * (I + 2) / 2 <= wv) I++;
    const int J = wv - I * (I + 1) / 2;
    const int i = lane & 15, kq = lane >> 4;
    for (int l = t; l < Lw; l += nthr) { sM[l] = d.lm_meta[l0 + l]; sCG[l] = d.lm_cg[l0 + l]; }
    __syncthreads();
    // panel element e of a pass: row r = e / ld (landmark lb + r), column c = e % ld
    double pf[R1_PF];
    auto fetch = [&](int lb) {
#pragma unroll
        for (int u2 = 0; u2 < R1_PF; u2++) {
            const int e = t + u2 * nthr, r = e / ld, c = e - r * ld, l = lb - l0 + r;
            double v = 0.0;
            if (e < R1_CHUNK * ld && l < Lw) {
                const unsigned m0 = sM[l];
                const int h6 = 6 * (int)(m0 & 255), k6 = 6 * (int)((m0 >> 8) & 255);
                if (c >= h6 && c < h6 + k6) v = d.W[(size_t)(fw0 + (int)(m0 >> 16) + l0 + l) * 6 + (c - h6)];
                else if (c == n6) v = sCG[l].y;
            }
            pf[u2] = v;
        }
    };
    double4v acc = {0, 0, 0, 0};
    fetch(l0);
    for (int lb = l0; lb < l1; lb += R1_CHUNK) {
        __syncthreads();                                   // the previous pass has been consumed
#pragma unroll
        for (int u2 = 0; u2 < R1_PF; u2++) {
            const int e = t + u2 * nthr, r = e / ld, c = e - r * ld;
            if (e < R1_CHUNK * ld) sW[r * lds_ld + c] = pf[u2];
        }
        __syncthreads();
        if (lb + R1_CHUNK < l1) fetch(lb + R1_CHUNK);      // in flight during the MFMAs below
#pragma unroll 4
        for (int k4 = 0; k4 < R1_CHUNK; k4 += 4) {
            const int l = k4 + kq, lg = lb - l0 + l;
            const double cl = lg < Lw ? sCG[lg].x : 0.0;
            const double av = sW[l * lds_ld + 16 * I + i] * cl, bv = sW[l * lds_ld + 16 * J + i];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
    }
    // C/D layout of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int R = 16 * I + kq + 4 * reg, Cc = 16 * J + i;
        if (R < n6 && Cc < n6 && R >= Cc) {
            const int fa = Cc / 6, c = Cc - 6 * fa, fb = R / 6, r = R - 6 * fb, bo = fb - fa;
            if (bo > 0 || c <= r) out[tvis_col(fa, N) + bo * 36 + r * 6 + c] -= acc[reg];
        } else if (R == n6 && Cc < n6) {
            out[tail + 12 * N + Cc] = -acc[reg];           // reduced right-hand side bs = -sum c_l g_l w_l
        }
    }
}
template __global__ void k_rank1_mfma<1>(DevBatch);
template __global__ void k_rank1_mfma<2>(DevBatch);
template __global__ void k_rank1_mfma<3>(DevBatch);
template __global__ void k_rank1_mfma<4>(DevBatch);
template __global__ void k_rank1_mfma<5>(DevBatch);

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
