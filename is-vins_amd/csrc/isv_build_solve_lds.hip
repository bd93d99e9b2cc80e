// isv_build_solve_lds.hip -- k_build_solve_lds: the tuned LDS-resident variant of k_build_solve for
// windows whose reduced system fits in LDS (15 N <= 165, i.e. N <= 11: BASELINE configs 2 and 4).
// Same mathematics and outputs as isv_build_solve.hip (see the header there); the differences are
// about the machine only:
//   * 768 threads = 12 wavefronts.  Wavefront a < N owns block column a (frame a) of the reduced
//     matrix T and keeps its <= 7 entries per lane in REGISTERS over the whole landmark sweep
//     (owner-computes, fixed landmark order => bitwise reproducible, no atomics, no LDS
//     read-modify-write).  T itself is only materialised in LDS after the sweep, so during the sweep
//     the 160 KB of LDS hold two 192-factor strip buffers: wavefront 11 is a dedicated LOADER that
//     streams the next chunk HBM -> registers -> LDS while the others consume the current one.
//   * barriers inside the sweep are raw s_barrier + lgkmcnt(0): they do not drain the loader's
//     outstanding global loads (a __syncthreads() would).
//   * frames covered by a landmark are found with a wave ballot over the chunk metadata; only
//     covering landmarks are visited, in index order.
//   * IMU / prior factors arrive as precomputed J^T J blocks (imu_H / prior_H).
//   * the 15x15 diagonal Cholesky blocks are factored AND inverted by one wavefront in registers
//     (v_readlane broadcasts, rsqrt + Newton instead of IEEE sqrt/div); with the block inverses the
//     panel solve and both triangular solves are mat-vecs.
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"

#define LT 768                     // threads
#define CHB 192                    // factors per staged chunk (host builds ck_rec with this bound)
#define MAXE 7                     // entries of a block column per lane: ceil(11 * 36 / 64)
#define MAXCK 64                   // chunk records cached in LDS

DEV int tblk(int I, int J) { return (I * (I + 1) / 2 + J) * 225; }
DEV int pairidx(int a, int b) { return a * (a + 1) / 2 + b; }      // a >= b

DEV double readlane_d(double v, int lane) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
// 1/sqrt(x) to ~1 ulp: hardware estimate + two Newton steps
DEV double rsqrt_nr(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
// LDS-only barrier: does not wait for outstanding global loads/stores
#define SYNC_LDS() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); } while (0)

#ifdef ISV_STAMP
#define STAMP(k) do { if (t == 0) { unsigned long long now_ = wall_clock64(); d.dbg[(size_t)w * 64 + (k)] += (double)(now_ - t_last); t_last = now_; } } while (0)
#define STAMPW(k) do { if (t == 4 * 64) { unsigned long long now_ = wall_clock64(); d.dbg[(size_t)w * 64 + (k)] += (double)(now_ - tw_last); tw_last = now_; } } while (0)
#else
#define STAMP(k) do {} while (0)
#define STAMPW(k) do {} while (0)
#endif

__global__ __launch_bounds__(LT) void k_build_solve_lds(DevBatch d) {
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, n = 15 * N, nblkT = N * (N + 1) / 2 * 225;
    double *p = lds;
    double *g = p; p += n;
    double *bs = p; p += n;
    double *hdiag = p; p += n;
    double *sc = p; p += n;
    double *D = p; p += n;
    double *y = p; p += n;
    double *u = p; p += n;
    double *red = p; p += LT;
    int2 *ckL = (int2 *)p; p += MAXCK + 1;
    int *flag = (int *)p; p += 1;
    if ((p - lds) & 1) p += 1;             // keep the strip buffers 16-byte aligned (ds_read_b128)
    double *big = p;
    // after the sweep: T and the block inverses
    double *T = big;
    double *LinvAll = big + nblkT;
    // during the sweep: two strip buffers, per-chunk landmark data
    double *sS0 = big;
    double *Wc = sS0 + 2 * CHB * 28;
    double *cC = Wc + 2 * CHB * 6;
    double *cG = cC + CHB;
    double *cSl = cG + CHB;                   // [2][CHB] Jacobi scales of the chunk (iteration > 0)
    unsigned *cM = (unsigned *)(cSl + 2 * CHB);   // [2][CHB] landmark metadata

#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64(), tw_last = wall_clock64();
#endif
    const int iteration = st.iteration;
    double mu = st.mu;
    int ls_fail = 0;
    double gmax_l = 0.0;
    const int ck0 = d.ck_off[w] + w, nchunks = d.ck_off[w + 1] - d.ck_off[w];
    const int fw0 = d.f_off[w];
    const int a = wv;                          // frame owned by this wavefront (a < N), wave 11 = loader
    const bool loader = (wv == LT / 64 - 1);

    for (int c = t; c <= nchunks && c <= MAXCK; c += LT) ckL[c] = d.ck_rec[ck0 + c];
    if (t == 0) flag[0] = 0;
    __syncthreads();

    int attempt = 0;                           // attempt 0 consumes k_sweep's result; a mu retry re-sweeps in-kernel
    for (;;) {
        if (!(mu < 1.0)) { ls_fail = 1; break; }
        for (int e = t; e < n; e += LT) { g[e] = 0.0; bs[e] = 0.0; hdiag[e] = 0.0; }
        // ---- P2 (retry path only): reprojection strips -----------------------------------------------
        double acc[12], hd = 0, gacc = 0, bacc = 0;     // acc[6*step + c]: row (bo, r) of block column a
#pragma unroll
        for (int i = 0; i < 12; i++) acc[i] = 0;
        auto load_chunk = [&](int c) {         // executed by the loader wavefront only
            const int2 r0 = ckL[c], r1 = ckL[c + 1];
            const int nf = r1.y - r0.y, nlm = r1.x - r0.x, tot = nf * ISV_PROJ_STRIP;
            const double *src = d.strip + (size_t)r0.y * ISV_PROJ_STRIP;
            double *dst = sS0 + (size_t)(c & 1) * CHB * 28;
            for (int base = 0; base < tot; base += 64 * 8) {
                double v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) { const int e = base + lane + 64 * i; v[i] = (e < tot) ? src[e] : 0.0; }
#pragma unroll
                for (int i = 0; i < 8; i++) { const int e = base + lane + 64 * i; if (e < tot) dst[e] = v[i]; }
            }
            for (int li = lane; li < nlm; li += 64) {
                cM[(c & 1) * CHB + li] = d.lm_meta[r0.x + li];
                cSl[(c & 1) * CHB + li] = (iteration == 0) ? 0.0 : d.scale_l[r0.x + li];
            }
        };
        const int nck = (attempt > 0) ? nchunks : 0;
        if (loader && nck > 0) load_chunk(0);
        STAMPW(13);
        for (int c = 0; c < nck; c++) {
            SYNC_LDS();                                        // chunk c staged; chunk c-1 fully consumed
            STAMPW(12);
            const int2 r0 = ckL[c], r1 = ckL[c + 1];
            const int clb = r0.x, cnlm = r1.x - r0.x, cnf = r1.y - r0.y, cfb = r0.y - fw0;
            const double *sS = sS0 + (size_t)(c & 1) * CHB * 28;
            const unsigned *cMc = cM + (c & 1) * CHB;
            if (loader) {
                if (c + 1 < nck) load_chunk(c + 1);            // streams while the others compute
            } else if (t < cnlm) {
                // (a1) per-landmark scalars and the host-frame w
                const unsigned m0 = cMc[t]; const int k = (m0 >> 8) & 255, fo = (int)(m0 >> 16) - cfb;
                double E = 0, gl = 0, wh[6] = {0, 0, 0, 0, 0, 0};
                for (int m = 0; m < k - 1; m++) {
                    const double *s = sS + (fo + m) * ISV_PROJ_STRIP;
                    const double j0 = s[26], j1 = s[27];
                    E += j0 * j0 + j1 * j1; gl += j0 * s[0] + j1 * s[1];
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) wh[cc] += s[2 + cc] * j0 + s[8 + cc] * j1;
                }
                const int l = clb + t;
                double sl;
                if (iteration == 0) { sl = 1.0 / (1.0 + sqrt(E)); d.scale_l[l] = sl; }
                else sl = cSl[(c & 1) * CHB + t];
                const double Es = sl * sl * E;
                const double Dl2 = fmin(fmax(Es, 1e-6), 1e32);
                const double Dl = sqrt(Dl2);
                cC[t] = sl * sl / (Es + mu * Dl2);
                cG[t] = gl;
                d.lmE[l] = E; d.lmG[l] = gl; d.diag_l[l] = Dl; d.grad_l[l] = sl * gl / Dl;
                gmax_l = fmax(gmax_l, fabs(gl));
                double *wo = Wc + (size_t)(fo + t) * 6;
#pragma unroll
                for (int cc = 0; cc < 6; cc++) wo[cc] = wh[cc];
            } else if (t >= 256 && t < 256 + cnf) {
                // (a2) per-factor w of the observing frame; landmark of the factor by bisection
                const int ff = t - 256;
                int lo = 0, hi = cnlm;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)(cMc[mid] >> 16) - cfb <= ff) lo = mid; else hi = mid; }
                const int li = lo, fo = (int)(cMc[li] >> 16) - cfb, m = ff - fo;
                const double *s = sS + ff * ISV_PROJ_STRIP;
                const double j0 = s[26], j1 = s[27];
                double *wo = Wc + (size_t)(fo + li + m + 1) * 6;
#pragma unroll
                for (int cc = 0; cc < 6; cc++) wo[cc] = s[14 + cc] * j0 + s[20 + cc] * j1;
            }
            SYNC_LDS();
            STAMPW(10);
            // (b) block column a: visit the landmarks of the chunk that cover frame a, in order.
            // Lane = (block row bo, row r) of the column: 6 entries (c = 0..5) per lane in registers.
            if (a < N) {
                for (int base = 0; base < cnlm; base += 64) {
                    const unsigned mm = (base + lane < cnlm) ? cMc[base + lane] : 0u;
                    const int hh = mm & 255, kk = (mm >> 8) & 255;
                    unsigned long long mask = __ballot(base + lane < cnlm && a >= hh && a < hh + kk);
                    while (mask) {
                        const int bit = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const unsigned m0 = __builtin_amdgcn_readlane(mm, bit);
                        const int li = base + bit, h = m0 & 255, k = (m0 >> 8) & 255;
                        const int fo = (int)(m0 >> 16) - cfb, pa = a - h, nb = k - pa;
                        const double cl = cC[li];
                        const double *wa = Wc + (size_t)(fo + li + pa) * 6;
                        const double2 wa01 = *reinterpret_cast<const double2 *>(wa), wa23 = *reinterpret_cast<const double2 *>(wa + 2),
                                      wa45 = *reinterpret_cast<const double2 *>(wa + 4);
#pragma unroll
                        for (int step = 0; step < 2; step++) {
                            if (step * 64 < 6 * nb) {                       // wave-uniform; step 1 only for 11-frame tracks
                                const int ll = lane + 64 * step, bo = ll / 6, r = ll - 6 * bo;
                                if (bo < nb) {
                                    const double wbr = wa[bo * 6 + r];
                                    const double coef = -cl * wbr;
                                    double *ac = acc + 6 * step;
                                    ac[0] += coef * wa01.x; ac[1] += coef * wa01.y; ac[2] += coef * wa23.x;
                                    ac[3] += coef * wa23.y; ac[4] += coef * wa45.x; ac[5] += coef * wa45.y;
                                    if (pa == 0) {
                                        // host column: factor m feeds block (h,h) [lanes bo == 0] and block (h+m+1, h) [lanes bo == m+1]
                                        for (int m = 0; m < k - 1; m++) {
                                            if (bo == 0 || bo == m + 1) {
                                                const double *s = sS + (fo + m) * ISV_PROJ_STRIP;
                                                const int ro = (bo == 0) ? 2 : 14;
                                                const double jr0 = s[ro + r], jr1 = s[ro + 6 + r];
                                                const double2 c01 = *reinterpret_cast<const double2 *>(s + 2), c23 = *reinterpret_cast<const double2 *>(s + 4),
                                                              c45 = *reinterpret_cast<const double2 *>(s + 6), d01 = *reinterpret_cast<const double2 *>(s + 8),
                                                              d23 = *reinterpret_cast<const double2 *>(s + 10), d45 = *reinterpret_cast<const double2 *>(s + 12);
                                                ac[0] += jr0 * c01.x + jr1 * d01.x; ac[1] += jr0 * c01.y + jr1 * d01.y;
                                                ac[2] += jr0 * c23.x + jr1 * d23.x; ac[3] += jr0 * c23.y + jr1 * d23.y;
                                                ac[4] += jr0 * c45.x + jr1 * d45.x; ac[5] += jr0 * c45.y + jr1 * d45.y;
                                                if (bo == 0) {
                                                    const double2 rs = *reinterpret_cast<const double2 *>(s);
                                                    hd += jr0 * jr0 + jr1 * jr1;
                                                    gacc += jr0 * rs.x + jr1 * rs.y;
                                                }
                                            }
                                        }
                                    } else if (bo == 0) {
                                        const double *s = sS + (fo + pa - 1) * ISV_PROJ_STRIP;
                                        const double jr0 = s[14 + r], jr1 = s[20 + r];
                                        const double2 c01 = *reinterpret_cast<const double2 *>(s + 14), c23 = *reinterpret_cast<const double2 *>(s + 16),
                                                      c45 = *reinterpret_cast<const double2 *>(s + 18), d01 = *reinterpret_cast<const double2 *>(s + 20),
                                                      d23 = *reinterpret_cast<const double2 *>(s + 22), d45 = *reinterpret_cast<const double2 *>(s + 24);
                                        const double2 rs = *reinterpret_cast<const double2 *>(s);
                                        ac[0] += jr0 * c01.x + jr1 * d01.x; ac[1] += jr0 * c01.y + jr1 * d01.y;
                                        ac[2] += jr0 * c23.x + jr1 * d23.x; ac[3] += jr0 * c23.y + jr1 * d23.y;
                                        ac[4] += jr0 * c45.x + jr1 * d45.x; ac[5] += jr0 * c45.y + jr1 * d45.y;
                                        hd += jr0 * jr0 + jr1 * jr1;
                                        gacc += jr0 * rs.x + jr1 * rs.y;
                                    }
                                    if (bo == 0) bacc -= cl * wbr * cG[li];
                                }
                            }
                        }
                    }
                }
            }
            STAMPW(11);
        }
        __syncthreads();
        STAMP(2);
        // ---- materialise T: zero, IMU band, priors, then the register accumulators ------------------
        for (int e = t; e < nblkT; e += LT) T[e] = 0.0;
        __syncthreads();
        {
            const double *H = d.imu_H + (size_t)w * (N - 1) * ISV_IMU_H;
            const int *skip = d.imu_skip + (size_t)w * (N - 1);
            const int nd = N * 120, no = (N - 1) * 225;
            for (int e = t; e < nd + no + n; e += LT) {
                if (e < nd) {
                    const int I = e / 120, pq = e % 120;
                    int r = 0;
                    while ((r + 1) * (r + 2) / 2 <= pq) r++;
                    const int c = pq - r * (r + 1) / 2;
                    double v = 0;
                    if (I >= 1 && !skip[I - 1]) v += H[(size_t)(I - 1) * ISV_IMU_H + pairidx(15 + r, 15 + c)];
                    if (I <= N - 2 && !skip[I]) v += H[(size_t)I * ISV_IMU_H + pairidx(r, c)];
                    T[tblk(I, I) + r * 15 + c] = v;
                    if (r == c) hdiag[15 * I + r] = v;
                } else if (e < nd + no) {
                    const int q = e - nd, I = q / 225, rc = q % 225, r = rc / 15, c = rc % 15;
                    if (!skip[I]) T[tblk(I + 1, I) + rc] = H[(size_t)I * ISV_IMU_H + pairidx(15 + r, c)];
                } else {
                    const int gi = e - nd - no, I = gi / 15, r = gi % 15;
                    double v = 0;
                    if (I >= 1 && !skip[I - 1]) v += H[(size_t)(I - 1) * ISV_IMU_H + 465 + 15 + r];
                    if (I <= N - 2 && !skip[I]) v += H[(size_t)I * ISV_IMU_H + 465 + r];
                    g[gi] = v;
                }
            }
            __syncthreads();
            const double *PH = d.prior_H + (size_t)w * d.prior_H_sz;
            const int nprior = 2 + (d.Nvo - 1) + d.n_rp[w];
            for (int q = 0; q < nprior; q++) {
                int ncol, off, c0, c1 = 0;
                if (q == 0) { ncol = 6; off = PH_SE3; c0 = 0; }
                else if (q == 1) { ncol = 9; off = PH_LIN9; c0 = 15 * (d.Nvo - 1) + 6; }
                else if (q < 1 + d.Nvo) { const int k = q - 2; ncol = 12; off = PH_REL0 + PH_REL_SZ * k; c0 = 15 * k; c1 = 15 * (k + 1); }
                else { const int m = q - 1 - d.Nvo; ncol = 6; off = PH_REL0 + PH_REL_SZ * (d.Nvo - 1) + PH_RP_SZ * m; c0 = 15 * d.rollpitch[(size_t)w * d.max_rp + m].index; }
                const int np2 = ncol * (ncol + 1) / 2;
                for (int e = t; e < np2 + ncol; e += LT) {
                    if (e < np2) {
                        int aa = 0;
                        while ((aa + 1) * (aa + 2) / 2 <= e) aa++;
                        const int bb = e - aa * (aa + 1) / 2;
                        const int ga = (aa < 6 || ncol != 12) ? c0 + aa : c1 + aa - 6;
                        const int gb = (bb < 6 || ncol != 12) ? c0 + bb : c1 + bb - 6;
                        const double v = PH[off + e];
                        T[tblk(ga / 15, gb / 15) + (ga % 15) * 15 + (gb % 15)] += v;
                        if (aa == bb) hdiag[ga] += v;
                    } else {
                        const int aa = e - np2;
                        const int ga = (aa < 6 || ncol != 12) ? c0 + aa : c1 + aa - 6;
                        g[ga] += PH[off + e];
                    }
                }
                __syncthreads();
            }
        }
        if (attempt == 0) {
            // reprojection part from k_sweep (6x6 pose corners of every block, owner = this thread)
            const double *V = d.Tvis + (size_t)w * d.tvis_sz;
            const int nblk36 = N * (N + 1) / 2 * 36;
            for (int e = t; e < nblk36; e += LT) {
                const int q = e / 36, rc = e - 36 * q, r = rc / 6, c = rc - 6 * r;
                int ca = 0;
                while (ca + 1 < N && (ca + 1) * N - (ca + 1) * ca / 2 <= q) ca++;
                const int bo = q - (ca * N - ca * (ca - 1) / 2);
                if (bo > 0 || c <= r) T[tblk(ca + bo, ca) + r * 15 + c] += V[e];
            }
            for (int e = t; e < 6 * N; e += LT) {
                const int fa = e / 6, r = e - 6 * fa;
                hdiag[15 * fa + r] += V[nblk36 + e]; g[15 * fa + r] += V[nblk36 + 6 * N + e]; bs[15 * fa + r] += V[nblk36 + 12 * N + e];
            }
            const int l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
            for (int l = l0 + t; l < l1; l += LT) gmax_l = fmax(gmax_l, fabs(d.lmG[l]));
        } else if (a < N) {
#pragma unroll
            for (int step = 0; step < 2; step++) {
                const int ll = lane + 64 * step, bo = ll / 6, r = ll - 6 * bo;
                if (a + bo < N) {
                    double *Tr = T + tblk(a + bo, a) + r * 15;
#pragma unroll
                    for (int c = 0; c < 6; c++) if (bo > 0 || c <= r) Tr[c] += acc[6 * step + c];
                    if (bo == 0) { hdiag[15 * a + r] += hd; g[15 * a + r] += gacc; bs[15 * a + r] += bacc; }
                }
            }
        }
        __syncthreads();
        STAMP(1);
        // ---- P3: scaling, Cauchy data ----------------------------------------------------------
        for (int e = t; e < n; e += LT) {
            double s;
            if (iteration == 0) { s = 1.0 / (1.0 + sqrt(hdiag[e])); d.scale_p[(size_t)w * n + e] = s; }
            else s = d.scale_p[(size_t)w * n + e];
            const double D2 = fmin(fmax(s * s * hdiag[e], 1e-6), 1e32);
            sc[e] = s; D[e] = sqrt(D2);
            d.diag_p[(size_t)w * n + e] = D[e];
            d.grad_p[(size_t)w * n + e] = s * g[e] / D[e];
            u[e] = s * s * g[e] / D2;
            d.up[(size_t)w * n + e] = u[e];
            y[e] = s * (g[e] + bs[e]);
        }
        __syncthreads();
        {   // qT = u^T T u on the unscaled T, then scale in place and add the LM diagonal
            double accq = 0;
            for (int e = t; e < nblkT; e += LT) {
                const int bq = e / 225, rc = e - 225 * bq, r = rc / 15, c = rc - 15 * r;
                int I = 0;
                while ((I + 1) * (I + 2) / 2 <= bq) I++;
                const int J = bq - I * (I + 1) / 2;
                const int gi = 15 * I + r, gj = 15 * J + c;
                const double v = T[e];
                if (I == J) { if (r > c) accq += 2.0 * v * u[gi] * u[gj]; else if (r == c) accq += v * u[gi] * u[gi]; }
                else accq += 2.0 * v * u[gi] * u[gj];
                double sv = v * sc[gi] * sc[gj];
                if (gi == gj) sv += mu * D[gi] * D[gi];
                T[e] = sv;
            }
            red[t] = accq;
            __syncthreads();
            for (int off = 512; off > 0; off >>= 1) { if (t < off && t + off < LT) red[t] += red[t + off]; __syncthreads(); }
            if (t == 0) st.qT = red[0];
        }
        __syncthreads();
        STAMP(3);
        // ---- blocked right-looking Cholesky with look-ahead ----------------------------------------
        // Diagonal blocks are factored AND inverted by wavefront 0 in registers (v_readlane); with
        // look-ahead the factorisation of block J+1 runs while wavefronts 1..11 finish the trailing
        // update of block J (only block column J+1 of that update has to be done first).
        auto diag_block = [&](int J) {                            // wavefront 0 only
            double *Dj = T + tblk(J, J);
            double *Li = LinvAll + J * 225;
            double row[15], dinv[15];
#pragma unroll
            for (int k = 0; k < 15; k++) row[k] = (lane < 15) ? Dj[lane * 15 + k] : 0.0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < 15; j++) {
                double s = row[j];
#pragma unroll
                for (int k = 0; k < j; k++) s -= row[k] * readlane_d(row[k], j);
                const double sj = readlane_d(s, j);           // pivot
                if (!(sj > 0.0)) bad = true;
                dinv[j] = rsqrt_nr(sj);                        // 1 / L_jj  (wave-uniform)
                row[j] = (lane == j) ? sj * dinv[j] : s * dinv[j];
            }
            if (lane == 0 && bad) flag[0] = 1;
            if (lane < 15) {
#pragma unroll
                for (int k = 0; k < 15; k++) Dj[lane * 15 + k] = (k <= lane) ? row[k] : 0.0;
#pragma unroll
                for (int k = 0; k < 15; k++) if (lane == k) Li[k] = dinv[k];     // 1 / L_kk in the first row of the (later) inverse slot
            }
        };
        auto invert_block = [&](int J) {                          // one wavefront per block, after the factorisation
            const double *Dj = T + tblk(J, J);
            double *Li = LinvAll + J * 225;
            double row[15], dinv[15], x[15];
#pragma unroll
            for (int k = 0; k < 15; k++) { row[k] = (lane < 15) ? Dj[lane * 15 + k] : 0.0; dinv[k] = Li[k]; }
#pragma unroll
            for (int i = 0; i < 15; i++) {                         // lane c solves L x = e_c
                double s = (lane == i) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; k++) s -= readlane_d(row[k], i) * x[k];
                x[i] = s * dinv[i];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < 15) {
#pragma unroll
                for (int k = 0; k < 15; k++) Li[k * 15 + lane] = x[k];
            }
        };
        auto trailing_tile = [&](int J, int I, int K, int tt) {  // 3x3 tile tt of T[I,K] -= L[I,J] L[K,J]^T
            const int tr = (tt / 5) * 3, tc = (tt % 5) * 3;
            const double *LI = T + tblk(I, J) + tr * 15, *LK = T + tblk(K, J) + tc * 15;
            double a00 = 0, a01 = 0, a02 = 0, a10 = 0, a11 = 0, a12 = 0, a20 = 0, a21 = 0, a22 = 0;
#pragma unroll
            for (int k = 0; k < 15; k++) {
                const double x0 = LI[k], x1 = LI[15 + k], x2 = LI[30 + k];
                const double z0 = LK[k], z1 = LK[15 + k], z2 = LK[30 + k];
                a00 += x0 * z0; a01 += x0 * z1; a02 += x0 * z2;
                a10 += x1 * z0; a11 += x1 * z1; a12 += x1 * z2;
                a20 += x2 * z0; a21 += x2 * z1; a22 += x2 * z2;
            }
            double *C = T + tblk(I, K) + tr * 15 + tc;
            C[0] -= a00; C[1] -= a01; C[2] -= a02;
            C[15] -= a10; C[16] -= a11; C[17] -= a12;
            C[30] -= a20; C[31] -= a21; C[32] -= a22;
        };
        if (wv == 0) diag_block(0);
        __syncthreads();
        for (int J = 0; J < N && !flag[0]; J++) {
            const double *Li = LinvAll + J * 225;
            const int prow = (N - J - 1) * 15;
            const double *Ljj = T + tblk(J, J);
            for (int rr = t; rr < prow; rr += LT) {             // panel: solve x L_JJ^T = a per row
                const int I = J + 1 + rr / 15, r = rr % 15;
                double *A = T + tblk(I, J) + r * 15;
                double xv[15];
#pragma unroll
                for (int c = 0; c < 15; c++) {
                    double s2 = A[c];
#pragma unroll
                    for (int k = 0; k < c; k++) s2 -= xv[k] * Ljj[c * 15 + k];
                    xv[c] = s2 * Li[c];
                }
#pragma unroll
                for (int c = 0; c < 15; c++) A[c] = xv[c];
            }
            __syncthreads();
            const int m = N - J - 1;
            if (m == 0) break;
            // part 1: block column J+1 of the trailing update (all threads)
            for (int tile = t; tile < m * 25; tile += LT) trailing_tile(J, J + 1 + tile / 25, J + 1, tile % 25);
            __syncthreads();
            // part 2: wavefront 0 factors block J+1 while the others update the remaining columns
            if (wv == 0) diag_block(J + 1);
            else {
                const int nb2 = m * (m - 1) / 2;                // blocks (I, K) with I >= K >= J+2
                for (int tile = t - 64; tile < nb2 * 25; tile += LT - 64) {
                    const int q = tile / 25;
                    int ii = 0;
                    while ((ii + 1) * (ii + 2) / 2 <= q) ii++;
                    const int kk = q - ii * (ii + 1) / 2;
                    trailing_tile(J, J + 2 + ii, J + 2 + kk, tile - 25 * q);
                }
            }
            __syncthreads();
        }
        __syncthreads();
        if (!flag[0] && wv < N) invert_block(wv);               // all diagonal-block inverses in parallel
        __syncthreads();
        STAMP(4);
        if (flag[0]) {
            mu *= 10.0; attempt++;
            __syncthreads();
            if (t == 0) flag[0] = 0;
            __syncthreads();
            continue;
        }
        // ---- solve with the block inverses ---------------------------------------------------------
        for (int J = 0; J < N; J++) {                           // forward
            const double *Li = LinvAll + J * 225;
            double v = 0;
            if (t < 15) {
#pragma unroll
                for (int k = 0; k < 15; k++) v += Li[t * 15 + k] * y[15 * J + k];
            }
            __syncthreads();
            if (t < 15) y[15 * J + t] = v;
            __syncthreads();
            for (int rr = t; rr < (N - J - 1) * 15; rr += LT) {
                const int I = J + 1 + rr / 15, r = rr % 15;
                const double *A = T + tblk(I, J) + r * 15;
                double s = 0;
#pragma unroll
                for (int k = 0; k < 15; k++) s += A[k] * y[15 * J + k];
                y[15 * I + r] -= s;
            }
            __syncthreads();
        }
        for (int J = N - 1; J >= 0; J--) {                      // backward
            const double *Li = LinvAll + J * 225;
            const int rows = (N - J - 1) * 15;
            if (t < 240) {
                const int c = t % 15, part = t / 15;
                double s = 0;
                for (int rr = part; rr < rows; rr += 16) {
                    const int I = J + 1 + rr / 15, r = rr % 15;
                    s += T[tblk(I, J) + r * 15 + c] * y[15 * I + r];
                }
                red[t] = s;
            }
            __syncthreads();
            if (t < 15) {
                double s = 0;
                for (int part = 0; part < 16; part++) s += red[part * 15 + t];
                red[256 + t] = y[15 * J + t] - s;
            }
            __syncthreads();
            if (t < 15) {
                double v = 0;
#pragma unroll
                for (int k = 0; k < 15; k++) v += Li[k * 15 + t] * red[256 + k];
                y[15 * J + t] = v;
            }
            __syncthreads();
        }
        STAMP(5);
        break;
    }
    if (!ls_fail) {
        for (int e = t; e < n; e += LT) {
            d.zp[(size_t)w * n + e] = sc[e] * y[e];
            d.gn_p[(size_t)w * n + e] = -D[e] * y[e];
        }
    }
    {
        double m = gmax_l;
        for (int i = t; i < N; i += LT) {
            const double *x = d.pose + ((size_t)w * N + i) * 7;
            double ng[6], xp[7];
            for (int k = 0; k < 6; k++) ng[k] = -g[15 * i + k];
            pose_plus(x, ng, xp);
            for (int k = 0; k < 7; k++) m = fmax(m, fabs(x[k] - xp[k]));
            for (int k = 0; k < 9; k++) m = fmax(m, fabs(g[15 * i + 6 + k]));
        }
        red[t] = m;
        __syncthreads();
        for (int off = 512; off > 0; off >>= 1) { if (t < off && t + off < LT) red[t] = fmax(red[t], red[t + off]); __syncthreads(); }
    }
    if (t == 0) {
        st.gmax = red[0];
        st.mu = mu;
        st.ls_fail = ls_fail;
        st.need_linearize = 0;
        st.fresh = 1;
        st.x_cost = d.cost[w];
        if (iteration == 0) {
            st.initial_cost = d.cost[w];
            d.trace_cost[(size_t)w * ISV_MAX_TRACE] = d.cost[w];
            d.trace_radius[(size_t)w * ISV_MAX_TRACE] = st.radius;
        }
        if (st.gmax <= 1e-10) st.termination = ISV_TERM_GRADIENT_TOL;
    }
    STAMP(6);
}

size_t build_solve_lds2_bytes(int N) {
    const size_t n = 15 * (size_t)N, nblkT = (size_t)N * (N + 1) / 2 * 225;
    size_t sweep = 2 * CHB * 28 + 2 * CHB * 6 + 2 * CHB + 2 * CHB + CHB;      // strips x2, W, cC, cG, cSl x2, cM x2 (uint)
    size_t solve = nblkT + (size_t)N * 225;
    size_t big = sweep > solve ? sweep : solve;
    return (7 * n + LT + (MAXCK + 1) + 2 + big + 8) * sizeof(double);
}
int build_solve_lds2_chunk() { return CHB; }
