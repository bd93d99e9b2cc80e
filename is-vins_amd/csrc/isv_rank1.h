// isv_rank1.h -- the rank-1 downdates of the landmark elimination (k_rank1_mfma, see isv_sweep.hip) as a device routine shared by
// the kernels that run it: k_rank1_mfma / k_schur_split (isv_sweep.hip) and k_lin_gram_chain (isv_build_solve_sb.hip), where the
// workgroup that linearised a window goes straight on to its downdates.
#pragma once
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"

// Tvis layout of one window: block column a at 36 * (a N - a (a-1) / 2), block (a+bo, a) = 36 doubles
__host__ __device__ inline int tvis_col(int a, int N) { return 36 * (a * N - a * (a - 1) / 2); }
// k_rank1_mfma: T -= sum_l c_l w_l w_l^T,
// where row l of the panel P is the landmark's w vector dense over the 6N pose columns (zero where a frame
// does not see it) with g_l appended in column 6N.  One wavefront per 16x16 output tile (lower triangle,
// zero padded to 16s), v_mfma_f64_16x16x4: A = c_l * P[l][16I + i] for 4 landmarks, B = P[l][16J + j].
// The dense panels multiply structural zeros, but one MFMA replaces ~1000 scalar lane-FMAs with their
// address arithmetic (the scalar sweep was issue bound).  Row 6N of the product is sum c_l g_l w_l = -bs,
// the reduced right-hand side.  The packed w vectors (HBM) are expanded to panel rows in LDS, 64 landmarks
// per pass; the loads of the next pass are in flight while the current one is multiplied.
typedef double double4v __attribute__((ext_vector_type(4)));
// R1_CHUNK = landmarks staged per pass (R1_CHUNK x wd_ld doubles of LDS)
// NT = panel width / 16, TPW = output tiles per wavefront (compile time: cheap index arithmetic, right-sized
// prefetch registers; TPW > 1 keeps the workgroup within 1024 threads for long windows)
// R1_CHUNK / MINW: pass size and minimum waves per SIMD.  Variants with fewer, fatter wavefronts and four workgroups per
// CU (<5, 4, 32, 4>, <5, 2, 32, 8>: a 1024-window launch in one round) measured 138 / 143 us against 96-100 us for one
// wavefront per tile: the per-workgroup MFMA chain gets longer than the round it saves.
// SPLIT (round 4): ONE window's landmarks over Gr workgroups (grid (B, Gs + Gr), this is group blockIdx.y - Gs): the group takes
// the passes [g P / Gr, (g + 1) P / Gr) of the window's P = ceil(L / R1_CHUNK) passes, forms the landmark scalars of exactly
// those landmarks, and leaves its raw accumulator tiles in d.r1_part for k_schur_fold (fixed order, no atomics).
// groups of a window with P passes: a function of the window alone (its bits do not depend on the batch around it)
// (round 5) ... and of the HANDLE's cap (DevBatch::split_cap: the groups per window that still give every workgroup of a full batch a
// resident slot, chosen from max_batch at creation): min(passes, ISV_SPLIT_MAX_GROUPS, cap), one group below two
__host__ __device__ inline int schur_split_groups(int P, int cap) {
    int g = P < ISV_SPLIT_MAX_GROUPS ? P : ISV_SPLIT_MAX_GROUPS;
    if (g > cap) g = cap;
    return (P >= ISV_SPLIT_MIN_PASSES && g >= 2) ? g : 1;
}
template <int NT, int TPW, int R1_CHUNK, int MINW, bool EX, bool SPLIT = false>
// (NT = 5, the benchmark's 11 frames: 15 wavefronts per workgroup, and two workgroups share a CU only at <= 64 VGPRs)
__device__ __forceinline__ void rank1_body(DevBatch &d, const int grp = 0, const int GrMax = 1) {
    constexpr int ntiles = NT * (NT + 1) / 2, nwaves = (ntiles + TPW - 1) / TPW;
    constexpr int ld = 16 * NT, nthr = 64 * nwaves;
    constexpr int R1_PF = (R1_CHUNK * ld + nthr - 1) / nthr;   // panel elements per thread and pass
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, n6 = 6 * N;
    int l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
    if (SPLIT) {                           // this group's passes: everything below indexes relative to ITS first landmark
        const int P = (l1 - l0 + R1_CHUNK - 1) / R1_CHUNK, Gr = schur_split_groups(P, d.split_cap);
        if (grp >= Gr) return;             // (uniform over the workgroup; a short window of the batch is one group: the unsplit sums)
        const int pa = (int)((long long)grp * P / Gr), pb = (int)((long long)(grp + 1) * P / Gr);
        const int e1 = l0 + R1_CHUNK * pb;
        l0 += R1_CHUNK * pa; l1 = e1 < l1 ? e1 : l1;
        if (l1 < l0) l1 = l0;
    }
    const int Lw = l1 - l0;
    constexpr int lds_ld = ld + 4;         // padded rows: the 4 k-rows of an operand hit distinct banks
    double *sW = lds;                      // [R1_CHUNK][ld + 4]
    double2 *sCG = (double2 *)(lds + R1_CHUNK * lds_ld);       // [max_lm] {c_l, g_l}
    unsigned *sM = (unsigned *)(sCG + d.max_lm);               // [max_lm] landmark metadata
    const int fw0 = d.f_off[w];
    double *out = d.Tvis + (size_t)w * d.tvis_sz;
    const int tail = 36 * (N * (N + 1) / 2);
    // wavefront wv owns the output tiles wv * TPW .. (I, J), I >= J
    int TI[TPW], TJ[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        const int q = wv * TPW + j < ntiles ? wv * TPW + j : ntiles - 1;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= q) I++;
        TI[j] = I; TJ[j] = q - I * (I + 1) / 2;
    }
    const int i = lane & 15, kq = lane >> 4;
#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64();
#define R1STAMP(k) do { if (t == 0) { unsigned long long now_ = wall_clock64(); d.dbg[(size_t)w * 64 + (k)] += (double)(now_ - t_last); t_last = now_; } } while (0)
#else
#define R1STAMP(k) do {} while (0)
#endif
    // landmark metadata first: host | k << 8 | (first factor - f_off[w]) << 16 -- everything the prologue and the passes index with
    for (int l = t; l < Lw; l += nthr) sM[l] = d.lm_meta[l0 + l];
    __syncthreads();
    if (d.fused_visual) {
        // Landmark scalars (what SchurEliminator needs per e-block) from the per-factor pieces k_lin_gram left, summed in
        // the landmark's own factor order: E = J_l^T J_l, g_l = J_l^T r, host-frame w = sum J_i^T J_l.  (The unfused
        // path does this inside k_proj_linearize<0>, where a landmark's factors are adjacent lanes.)
        // four threads per landmark: part 0 sums {E, g} and forms the scalars, parts 1..3 sum one pair of the host w each.
        // The pieces of up to four factors are loaded TOGETHER (clamped addresses, masked adds: same order, same bits);
        // one load per loop trip cost a memory latency per factor of the track.
        for (int q = t; q < 4 * Lw; q += nthr) {
            const int l = q >> 2, part = q & 3, gl = l0 + l;
            const unsigned m0 = sM[l];
            const int kf = (int)((m0 >> 8) & 255) - 1, f0 = fw0 + (int)(m0 >> 16);
            const double2 *fl = (const double2 *)(d.flm + (size_t)f0 * 8) + part;
            double2 acc = fl[0];
            for (int o = 1; o < kf; o += 4) {
                double2 a[4];
#pragma unroll
                for (int j = 0; j < 4; j++) a[j] = fl[4 * (o + j < kf ? o + j : kf - 1)];
#pragma unroll
                for (int j = 0; j < 4; j++) if (o + j < kf) { acc.x += a[j].x; acc.y += a[j].y; }
            }
            if (part == 0) {
                double sl;
                if (st.iteration == 0) { sl = 1.0 / (1.0 + sqrt(acc.x)); d.scale_l[gl] = sl; }
                else sl = d.scale_l[gl];
                const double Es = sl * sl * acc.x;
                const double Dl2 = fmin(fmax(Es, 1e-6), 1e32);
                const double Dl = sqrt(Dl2);
                const double2 cg = make_double2(sl * sl / (Es + st.mu * Dl2), acc.y);
                d.lm_cg[gl] = cg; sCG[l] = cg;
                d.lmE[gl] = acc.x; d.lmG[gl] = acc.y; d.diag_l[gl] = Dl; d.grad_l[gl] = sl * acc.y / Dl;
            } else {
                ((double2 *)(d.W + (size_t)(f0 + gl) * 6))[part - 1] = acc;     // host observation slot
            }
        }
        if (EX) {                                    // the extrinsic's w of every landmark: sum over its factors of J_ex^T J_l
            for (int q = t; q < 3 * Lw; q += nthr) {
                const int l = q / 3, part = q - 3 * l, gl = l0 + l, kf = d.lm_k[gl] - 1;
                const double2 *fx = (const double2 *)(d.flmx + (size_t)d.lm_f0[gl] * 6) + part;
                double2 acc = fx[0];
                for (int o = 1; o < kf; o++) { const double2 a = fx[3 * o]; acc.x += a.x; acc.y += a.y; }
                ((double2 *)(d.Wex + (size_t)gl * 6))[part] = acc;
            }
        }
    } else {
        for (int l = t; l < Lw; l += nthr) sCG[l] = d.lm_cg[l0 + l];
    }
    __syncthreads();                                       // the host slots (global) and sCG are read below by other threads
    R1STAMP(10);
    // panel element e of a pass: row r = e / ld (landmark lb + r), column c = e % ld.
    // BRANCH-FREE (round 3): every element issues its (clamped, always valid) load unconditionally and selects afterwards.  Written
    // with `if`s the compiler kept each load inside its own branch with an s_waitcnt vmcnt(0) before the next one (the
    // destination register is cleared on the other path), so the R1_PF gathers of a pass ran one HBM / L2 latency after
    // the other -- and the "prefetch" of the next pass stalled before the MFMAs it was meant to hide behind.
    double pf[R1_PF];
    unsigned pfsel = 0;                                            // 2 bits per element: 0 zero, 1 the loaded w entry, 2 g_l (LDS), 3 the extrinsic's w (loaded)
    const double *Wwin = d.W + (size_t)(fw0 + l0) * 6;             // this window's packed w vectors
    // (when the workgroup size is a multiple of the panel width -- NT <= 5 -- a thread's elements share ONE column and their
    // rows step by nthr / ld: two loop-invariant registers instead of a hoisted (row, column) pair per element)
    constexpr bool r1_aligned = nthr % ld == 0;
    const int r_first = t / ld, c_first = t - r_first * ld;
    auto fetch = [&](int lb) {
        pfsel = 0;
#pragma unroll
        for (int u2 = 0; u2 < R1_PF; u2++) {
            const int e = t + u2 * nthr;
            const int r = r1_aligned ? r_first + u2 * (nthr / ld) : e / ld, c = r1_aligned ? c_first : e - r * ld, l = lb - l0 + r;
            const bool valid = e < R1_CHUNK * ld && l < Lw;
            const int lc = valid ? l : 0;                          // (Lw >= 1 inside the pass loop)
            const unsigned m0 = sM[lc];
            const int h6 = 6 * (int)(m0 & 255), k6 = 6 * (int)((m0 >> 8) & 255);
            const bool inw = valid && c >= h6 && c < h6 + k6;
            unsigned sel = inw ? 1u : ((valid && c == n6) ? 2u : 0u);
            if (EX) {                                              // pseudo-frame columns = the extrinsic block
                const bool inx = valid && !inw && c >= 6 * d.Nr && c < n6;
                const double *src = inx ? d.Wex + (size_t)(l0 + lc) * 6 + (c - 6 * d.Nr) : Wwin + (inw ? ((int)(m0 >> 16) + lc) * 6 + (c - h6) : 0);
                pf[u2] = *src;
                if (inx) sel = 3u;
            } else {
                // (a 32-bit byte offset from the uniform base: one address VGPR per gather instead of two -- the kernel must stay within 64)
                pf[u2] = *(const double *)((const char *)Wwin + (size_t)(unsigned)(inw ? (((int)(m0 >> 16) + lc) * 6 + (c - h6)) * 8 : 0));
            }
            pfsel |= sel << (2 * u2);
        }
    };
    // the selection happens when the pass is written to LDS, i.e. AFTER the MFMAs of the previous pass: the loads stay in flight meanwhile
    auto commit = [&](int lb) {
#pragma unroll
        for (int u2 = 0; u2 < R1_PF; u2++) {
            const int e = t + u2 * nthr;
            const int r = r1_aligned ? r_first + u2 * (nthr / ld) : e / ld, c = r1_aligned ? c_first : e - r * ld;
            const unsigned sel = (pfsel >> (2 * u2)) & 3u;
            double v = (sel & 1u) ? pf[u2] : 0.0;
            if (sel == 2u) v = sCG[lb - l0 + r].y;
            if (e < R1_CHUNK * ld) sW[r * lds_ld + c] = v;
        }
    };
    double4v acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; j++) acc[j] = double4v{0, 0, 0, 0};
    if (Lw > 0) fetch(l0);
    for (int lb = l0; lb < l1; lb += R1_CHUNK) {
        __syncthreads();                                   // the previous pass has been consumed
        R1STAMP(11);
        commit(lb);
        __syncthreads();
        R1STAMP(12);
        if (lb + R1_CHUNK < l1) fetch(lb + R1_CHUNK);      // in flight during the MFMAs below
        R1STAMP(13);
#pragma unroll 4
        for (int k4 = 0; k4 < R1_CHUNK; k4 += 4) {
            const int l = k4 + kq, lg = lb - l0 + l;
            const double cl = lg < Lw ? sCG[lg].x : 0.0;
#pragma unroll
            for (int j = 0; j < TPW; j++) {
                const double av = sW[l * lds_ld + 16 * TI[j] + i] * cl, bv = sW[l * lds_ld + 16 * TJ[j] + i];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[j], 0, 0, 0);
            }
        }
        R1STAMP(14);
    }
    if (SPLIT) {                           // raw accumulator tiles of this group: [w][group][tile][reg][lane], coalesced
        double *part = d.r1_part + ((size_t)w * GrMax + grp) * (size_t)(ntiles * 256);
#pragma unroll
        for (int j = 0; j < TPW; j++) {
            if (wv * TPW + j >= ntiles) continue;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) part[(wv * TPW + j) * 256 + reg * 64 + lane] = acc[j][reg];
        }
        return;
    }
    // C/D layout of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 * reg
    // (the four read-modify-writes of a tile: all loads first -- clamped, always valid -- then the stores)
#pragma unroll
    for (int j = 0; j < TPW; j++) {
        if (wv * TPW + j >= ntiles) continue;
        int off[4]; double cur[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int R = 16 * TI[j] + kq + 4 * reg, Cc = 16 * TJ[j] + i;
            off[reg] = -1;
            if (R < n6 && Cc < n6 && R >= Cc) {
                const int fa = Cc / 6, c = Cc - 6 * fa, fb = R / 6, r = R - 6 * fb, bo = fb - fa;
                if (bo > 0 || c <= r) off[reg] = tvis_col(fa, N) + bo * 36 + r * 6 + c;
            } else if (R == n6 && Cc < n6) {
                off[reg] = -2 - Cc;                        // reduced right-hand side bs = -sum c_l g_l w_l  (plain store)
            }
            cur[reg] = out[off[reg] >= 0 ? off[reg] : 0];
        }
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            if (off[reg] >= 0) out[off[reg]] = cur[reg] - acc[j][reg];
            else if (off[reg] <= -2) out[tail + 12 * N + (-2 - off[reg])] = -acc[j][reg];
        }
    }
    R1STAMP(15);
}
