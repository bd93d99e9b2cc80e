// isv_sweep.hip -- elimination of the landmarks of every window (the e-block work of ceres'
// SchurEliminator, internal/ceres/schur_eliminator_impl.h Eliminate()), as two FP64-MFMA kernels:
//   k_sweep_mfma   DIRECT part of the reduced camera matrix: sum over the reprojection factors of
//                  J_p^T J_p (pose blocks), their gradient and the Jacobi-scaling diagonal, as Gram products
//                  over (host, observer) frame-pair groups.
//   k_rank1_mfma   the rank-1 downdates  - c_l w_l w_l^T  and the reduced right-hand side, as dense panels.
// Both write the packed lower block triangle Tvis (6x6 pose blocks) that k_build_solve_sb assembles from.
// The landmark scalars (E_l, g_l, Jacobi scale, w of the host frame) come from k_proj_linearize<0>; the
// back-substitution lives in k_dogleg.
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"
#include "isv_rank1.h"


// DIRECT part as FP64 MFMA Gram products.  The factors of a window are sorted by (host h, observer j)
// frame pair at upload (pg_perm / pg_off).  For one pair the stacked rows X = [J_i | J_j | r] (2 rows
// per factor, 13 columns) give, in ONE accumulator tile G = X^T X (v_mfma_f64_16x16x4, two factors per
// instruction, A and B operands are the same register):
//     G[0:6, 0:6] = sum J_i^T J_i  (part of block (h,h))     G[6:12, 0:6]  = sum J_j^T J_i = block (j,h)
//     G[6:12,6:12] = sum J_j^T J_j (part of block (j,j))     G[0:6,12], G[6:12,12] = J_i^T r, J_j^T r
// The pair groups are spread over the ISV_SWEEP_WAVES wavefronts of the workgroup by a longest-first schedule built
// at upload (pg_sched); every group leaves its five pieces in LDS / Tvis, then the (a,a) blocks, the
// Jacobi diagonal and the gradient are folded in a fixed order (bitwise reproducible, no atomics).
// Every strip is read exactly once.
typedef double double4s __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64 * ISV_SWEEP_WAVES, 8) void k_sweep_mfma(DevBatch d) {
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, NP = N * (N - 1) / 2;
    // pair partials: in LDS while four workgroups still share a CU that way (NP * 84 doubles <= 40 KB), else in a global
    // scratch slice of this window (written and read back by this workgroup only: L2 traffic, one launch round kept;
    // chosen per launch for batches of more than one workgroup per CU -- a lone window is faster out of LDS)
    double *Pjj = d.sw_global ? d.sw_part + (size_t)w * NP * 84 : lds;      // [NP][36]  sum J_j^T J_j of pair p
    double *Phh = Pjj + NP * 36;           // [NP][36]  sum J_i^T J_i of pair p
    double *Pgj = Phh + NP * 36;           // [NP][6]   sum J_j^T r
    double *Pgh = Pgj + NP * 6;            // [NP][6]   sum J_i^T r
    int *offL = d.sw_global ? (int *)lds : (int *)(Pgh + NP * 6);     // [NP + 1] group starts, staged once
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
    // this wavefront's group records, 64 at a time (one per lane, broadcast by __shfl); a wavefront can own MORE than
    // 64 groups (N = 18: 153 pairs, most of them empty when the landmarks share few host frames), so the records are
    // reloaded at every 64-group boundary
    int myrec = (q0 + lane < q1) ? sched[q0 + lane] : 0;
    for (int e = t; e <= NP; e += blockDim.x) offL[e] = d.pg_off[(size_t)w * (NP + 1) + e];
    __syncthreads();
    SWSTAMP(40);
    // first chunk of the first group; the next group's chunk is prefetched while the current one is consumed
    int nxt = 0;
    if (q0 < q1) { const int p = __shfl(myrec, 0) >> 16; const int b0 = offL[p], c0 = offL[p + 1] - b0; nxt = lane < c0 ? perm[b0 + lane] : 0; }
    for (int q = q0; q < q1; q++) {
        const int rel = (q - q0) & 63;
        const int rec = __shfl(myrec, rel), h = rec & 255, j = (rec >> 8) & 255, p = rec >> 16;
        const int b0 = offL[p], b1 = offL[p + 1];
        int myf = nxt;
        if (rel == 63) myrec = (q + 1 + lane < q1) ? sched[q + 1 + lane] : 0;       // next 64 records (rec is already out)
        if (q + 1 < q1) { const int p2 = __shfl(myrec, (rel + 1) & 63) >> 16; const int b2 = offL[p2], c2 = offL[p2 + 1] - b2; nxt = lane < c2 ? perm[b2 + lane] : 0; }
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
                    // (round 4: unconditional, clamped loads; masked operand entries are multiplied by zero where they are used.
                    // As `cond ? load : 0.0` the compiler sank every load into a branch of its own behind an s_waitcnt vmcnt(0):
                    // eight serialised memory latencies per round instead of one)
                    v[u2] = strip[(size_t)f * ISV_PROJ_STRIP + (colok ? eoff : 0)];
                }
#pragma unroll
                for (int u2 = 0; u2 < 8; u2++)
                    if (s2 + 2 * u2 < cnt) {
                        const double x = v[u2] * ((colok && s2 + 2 * u2 + fsel < cnt) ? 1.0 : 0.0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
                    }
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
    for (int tq = t; tq < N * 42; tq += blockDim.x) {
        if (tq < N * 36) {                 // diagonal blocks and the Jacobi-scaling diagonal
            const int a = tq / 36, rc = tq - 36 * a, r = rc / 6, c = rc - 6 * r;
            double s = 0.0;
            for (int j2 = a + 1; j2 < N; j2++) s += Phh[pidx(a, j2) * 36 + rc];
            for (int h2 = 0; h2 < a; h2++) s += Pjj[pidx(h2, a) * 36 + rc];
            out[tvis_col(a, N) + rc] = s;
            if (r == c) out[tail + 6 * a + r] = s;
        } else {                           // gradient
            const int q = tq - N * 36, a = q / 6, r = q - 6 * a;
            double s = 0.0;
            for (int j2 = a + 1; j2 < N; j2++) s += Pgh[pidx(a, j2) * 6 + r];
            for (int h2 = 0; h2 < a; h2++) s += Pgj[pidx(h2, a) * 6 + r];
            out[tail + 6 * N + 6 * a + r] = s;
        }
    }
    SWSTAMP(43);
}

// Rank-1 landmark downdates as FP64 MFMA panels:  Tvis -= P^T diag(c) P  over the window's landmarks,
template <int NT, int TPW, int R1_CHUNK, int MINW, bool EX>
__global__ __launch_bounds__(64 * ((NT * (NT + 1) / 2 + TPW - 1) / TPW), MINW) void k_rank1_mfma(DevBatch d) { rank1_body<NT, TPW, R1_CHUNK, MINW, EX>(d); }
// NT = 5 (the benchmark's 11 frames): 15 wavefronts per workgroup, and two workgroups share a CU only at eight wavefronts
// per SIMD, i.e. <= 64 VGPRs -- the launch bounds alone let the compiler settle at 66-70 (occupancy 7: ONE workgroup per CU)
template <> __global__ __launch_bounds__(960) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_rank1_mfma<5, 1, 64, 1, false>(DevBatch d) { rank1_body<5, 1, 64, 1, false>(d); }
template __global__ void k_rank1_mfma<1, 1, 64, 1, false>(DevBatch);
template __global__ void k_rank1_mfma<1, 1, 64, 1, true>(DevBatch);
template __global__ void k_rank1_mfma<2, 1, 64, 1, false>(DevBatch);
template __global__ void k_rank1_mfma<2, 1, 64, 1, true>(DevBatch);
template __global__ void k_rank1_mfma<3, 1, 64, 1, false>(DevBatch);
template __global__ void k_rank1_mfma<3, 1, 64, 1, true>(DevBatch);
template __global__ void k_rank1_mfma<4, 1, 64, 1, false>(DevBatch);
template __global__ void k_rank1_mfma<4, 1, 64, 1, true>(DevBatch);
template __global__ void k_rank1_mfma<5, 1, 64, 1, true>(DevBatch);
template __global__ void k_rank1_mfma<6, 2, 64, 1, false>(DevBatch);
template __global__ void k_rank1_mfma<6, 2, 64, 1, true>(DevBatch);
template __global__ void k_rank1_mfma<7, 2, 64, 1, false>(DevBatch);
template __global__ void k_rank1_mfma<7, 2, 64, 1, true>(DevBatch);
template __global__ void k_rank1_mfma<8, 3, 64, 1, false>(DevBatch);
template __global__ void k_rank1_mfma<8, 3, 64, 1, true>(DevBatch);


// ------------------------------------------------------------------------------------------------------------------
// (round 4) ONE window over many compute units: the landmark elimination of a long window in a small batch (BASELINE
// config 5: one window of 2000 landmarks / 30 000 factors) used ONE workgroup -- one CU of 256 -- for each of its two
// kernels.  k_schur_split runs both parts side by side on Gs + Gr workgroups per window:
//   groups [0, Gs)       the DIRECT part (k_sweep_mfma's Gram products): the window's (host, observer) pair groups dealt
//                        longest-first over the Gs x nwaves wavefronts (rank by size in LDS, snake order); a pair group is
//                        still summed by ONE wavefront in its stored factor order, so every pair partial has the bits the
//                        one-workgroup kernel gives; partials -> d.sw_part, blocks (j, h) -> Tvis;
//   groups [Gs, Gs + Gr) the rank-1 downdates (rank1_body<SPLIT>): Gr landmark ranges, raw accumulator tiles -> d.r1_part.
// k_schur_fold then folds the pair partials into the diagonal blocks / Jacobi diagonal / gradient (k_sweep_mfma's own
// order) and subtracts the Gr tile partials in group order: fixed order, no atomics, run-to-run reproducible.  Gr depends
// on the window's landmark count only (isv_solver.hip: schur_split_groups), so a window has ONE split result whatever
// the batch around it; it differs from the one-workgroup result in the last bits of the downdate (partial sums).
DEV void sweep_split_body(DevBatch &d, const int grp, const int Gs, const int nwaves) {
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, NP = N * (N - 1) / 2, nthr = 64 * nwaves;
    int *offL = (int *)lds;                    // [NP + 1] group starts
    int *order = offL + NP + 1;                // [NP] pairs by descending size (ties: ascending pair index)
    double *part = d.sw_part + (size_t)w * NP * 84;
    double *Pjj = part, *Phh = Pjj + NP * 36, *Pgj = Phh + NP * 36, *Pgh = Pgj + NP * 6;
    const int *perm = d.pg_perm + d.f_off[w];
    const double *strip = d.strip + (size_t)d.f_off[w] * ISV_PROJ_STRIP;
    double *out = d.Tvis + (size_t)w * d.tvis_sz;
    for (int e = t; e <= NP; e += nthr) offL[e] = d.pg_off[(size_t)w * (NP + 1) + e];
    __syncthreads();
    for (int p = t; p < NP; p += nthr) {
        const int c = offL[p + 1] - offL[p];
        int r = 0;
        for (int q = 0; q < NP; q++) { const int cq = offL[q + 1] - offL[q]; r += (cq > c || (cq == c && q < p)) ? 1 : 0; }
        order[r] = p;
    }
    __syncthreads();
    const int TW = Gs * nwaves, gw = grp * nwaves + wv;
    const int i = lane & 15, kq = lane >> 4, row2 = kq & 1, fsel = kq >> 1;
    const int eoff = i < 6 ? 2 + row2 * 6 + i : (i < 12 ? 14 + row2 * 6 + (i - 6) : row2);   // strip element of operand column i
    const bool colok = i < 13;
    for (int k = 0; k * TW < NP; k++) {
        const int r = k * TW + ((k & 1) ? TW - 1 - gw : gw);
        if (r >= NP) continue;                 // (wave-uniform)
        const int p = order[r];
        int h = 0;
        while ((h + 1) * N - (h + 1) * (h + 2) / 2 <= p) h++;
        const int j = h + 1 + (p - (h * N - h * (h + 1) / 2));
        const int b0 = offL[p], b1 = offL[p + 1];
        double4s acc = {0, 0, 0, 0};
        // 16 factors (8 MFMAs) per round, four rounds per 64-factor chunk.  The gathers of chunk c + 1 are issued round by round
        // behind the MFMAs of chunk c (a register ring of four rounds), and the factor ids of chunk c + 2 (one coalesced load,
        // broadcast by __shfl) are requested before them: a wavefront keeps 32 strip loads in flight and never waits for an id
        // load with strip loads queued behind it.  (First version: ids fetched per round -- a dependent load whose s_waitcnt
        // drained the whole ring: two memory latencies per round, 60 us for the 268-factor pair groups of BASELINE config 5.)
        // Same MFMA order as k_sweep_mfma: same bits per pair group.
        const int nrounds = (b1 - b0 + 15) / 16, nch = (b1 - b0 + 63) / 64;
        auto ids = [&](int c) { const int pos = b0 + 64 * c + lane; return perm[pos < b1 ? pos : (b1 > b0 ? b1 - 1 : 0)]; };
        int myf0 = ids(0), myf1 = ids(1);
        double v[4][8];
        auto gather = [&](int q, int mf, double (&dst)[8]) {         // round q (ids of its chunk in mf); clamped, always valid addresses
#pragma unroll
            for (int u2 = 0; u2 < 8; u2++) {
                const int src = 16 * (q & 3) + 2 * u2 + fsel;
                const int f = __shfl(mf, src);
                // (masked operand entries are MULTIPLIED by zero after an unconditional, clamped load: with `cond ? load : 0.0` the
                // compiler sinks the load into a branch of its own with an s_waitcnt vmcnt(0) behind it -- eight serialised memory
                // latencies per round, 60 us per launch measured; the strips of valid factors are finite, so x * 0 is a zero)
                // (the ring keeps the RAW value: the multiplication happens where the operand is used, so the wait for a round's loads
                // sits in front of its MFMAs, four rounds later)
                dst[u2] = strip[(size_t)f * ISV_PROJ_STRIP + (colok ? eoff : 0)];
            }
        };
#pragma unroll
        for (int q = 0; q < 4; q++) if (q < nrounds) gather(q, myf0, v[q]);
        for (int c = 0; c < nch; c++) {
            const int myf2 = ids(c + 2);
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                const int q = 4 * c + qq;
                if (q < nrounds) {
                    const int left = b1 - b0 - 16 * q;                  // factors from this round on
#pragma unroll
                    for (int u2 = 0; u2 < 8; u2++)
                        if (2 * u2 < left) {
                            const double x = v[qq][u2] * ((colok && 2 * u2 + fsel < left) ? 1.0 : 0.0);
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
                        }
                    if (q + 4 < nrounds) gather(q + 4, myf1, v[qq]);
                }
            }
            myf0 = myf1; myf1 = myf2;
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
}

template <int NT, int TPW>
__global__ __launch_bounds__(64 * ((NT * (NT + 1) / 2 + TPW - 1) / TPW)) void k_schur_split(DevBatch d, int Gs, int GrMax) {
    constexpr int nwaves = (NT * (NT + 1) / 2 + TPW - 1) / TPW;
#ifdef ISV_SPLIT_SKIP          /* timing experiment: 1 = no direct part, 2 = no downdates (results are wrong) */
    if (ISV_SPLIT_SKIP == 1 && (int)blockIdx.y < Gs) return;
    if (ISV_SPLIT_SKIP == 2 && (int)blockIdx.y >= Gs) return;
#endif
    if ((int)blockIdx.y < Gs) sweep_split_body(d, blockIdx.y, Gs, nwaves);
    else rank1_body<NT, TPW, 64, 1, false, true>(d, (int)blockIdx.y - Gs, GrMax);
}
// the downdates alone (handles whose direct part k_lin_gram forms: Gs = 0): without the direct part's registers a workgroup stays at
// <= 64 VGPRs for the benchmark's 11 frames, two of them share a CU, and a batch of 128 windows can split every window four ways
template <int NT, int TPW>
__global__ __launch_bounds__(64 * ((NT * (NT + 1) / 2 + TPW - 1) / TPW)) void k_rank1_split(DevBatch d, int GrMax) { rank1_body<NT, TPW, 64, 1, false, true>(d, (int)blockIdx.y, GrMax); }
template <> __global__ __launch_bounds__(960) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_rank1_split<5, 1>(DevBatch d, int GrMax) { rank1_body<5, 1, 64, 1, false, true>(d, (int)blockIdx.y, GrMax); }
template __global__ void k_rank1_split<1, 1>(DevBatch, int);
template __global__ void k_rank1_split<2, 1>(DevBatch, int);
template __global__ void k_rank1_split<3, 1>(DevBatch, int);
template __global__ void k_rank1_split<4, 1>(DevBatch, int);
template __global__ void k_rank1_split<6, 2>(DevBatch, int);
template __global__ void k_rank1_split<7, 2>(DevBatch, int);
template __global__ void k_rank1_split<8, 3>(DevBatch, int);
template __global__ void k_schur_split<1, 1>(DevBatch, int, int);
template __global__ void k_schur_split<2, 1>(DevBatch, int, int);
template __global__ void k_schur_split<3, 1>(DevBatch, int, int);
template __global__ void k_schur_split<4, 1>(DevBatch, int, int);
template __global__ void k_schur_split<5, 1>(DevBatch, int, int);
template __global__ void k_schur_split<6, 2>(DevBatch, int, int);
template __global__ void k_schur_split<7, 2>(DevBatch, int, int);
template __global__ void k_schur_split<8, 3>(DevBatch, int, int);

// Fold of the split elimination (grid (B, any)): every thread owns whole output entries, so any number of workgroups works.
//   from_partials != 0 (the direct part was split too): the diagonal pose blocks, the Jacobi-scaling diagonal and the gradient
//       are the fixed-order sums over the pair partials (exactly k_sweep_mfma's fold);
//   then, per accumulator-tile entry: Tvis -= sum over the Gr group partials in group order; row 6N = the reduced rhs.
__global__ __launch_bounds__(256) void k_schur_fold(DevBatch d, int GrMax, int NT, int from_partials) {
    const int w = blockIdx.x;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, NP = N * (N - 1) / 2, n6 = 6 * N, ntiles = NT * (NT + 1) / 2;
    const int tid = blockIdx.y * 256 + threadIdx.x, nthr = gridDim.y * 256;
    double *out = d.Tvis + (size_t)w * d.tvis_sz;
    const double *part = d.sw_part ? d.sw_part + (size_t)w * NP * 84 : nullptr;
    const double *Pjj = part, *Phh = part + NP * 36, *Pgj = part + NP * 72, *Pgh = part + NP * 78;
    const int tail = 36 * (N * (N + 1) / 2);
    auto pidx = [N](int hh, int jj) { return hh * N - hh * (hh + 1) / 2 + (jj - hh - 1); };
    auto diag_entry = [&](int a, int rc) {             // entry rc of diagonal block a: pairs (a, j2) ascending, then (h2, a) ascending
        double s = 0.0;
#pragma unroll 4
        for (int j2 = a + 1; j2 < N; j2++) s += Phh[pidx(a, j2) * 36 + rc];
#pragma unroll 4
        for (int h2 = 0; h2 < a; h2++) s += Pjj[pidx(h2, a) * 36 + rc];
        return s;
    };
    if (from_partials) {
        for (int tq = tid; tq < N * 42; tq += nthr) {
            if (tq < N * 36) {
                const int a = tq / 36, rc = tq - 36 * a, r = rc / 6, c = rc - 6 * r;
                if (c < r) continue;                   // (the lower triangle is written below, together with its downdate)
                const double s = diag_entry(a, rc);
                if (c > r) out[tvis_col(a, N) + rc] = s;
                else out[tail + 6 * a + r] = s;        // Jacobi-scaling diagonal: the direct part, before the downdates
            } else {
                const int q = tq - N * 36, a = q / 6, r = q - 6 * a;
                double s = 0.0;
                for (int j2 = a + 1; j2 < N; j2++) s += Pgh[pidx(a, j2) * 6 + r];
                for (int h2 = 0; h2 < a; h2++) s += Pgj[pidx(h2, a) * 6 + r];
                out[tail + 6 * N + 6 * a + r] = s;
            }
        }
    }
    const int Gr = schur_split_groups((d.lm_off[w + 1] - d.lm_off[w] + 63) / 64, d.split_cap);
    const double *r1 = d.r1_part + (size_t)w * GrMax * (size_t)(ntiles * 256);
    for (int e = tid; e < ntiles * 256; e += nthr) {
        const int tile = e >> 8, reg = (e >> 6) & 3, lane = e & 63;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= tile) I++;
        const int J = tile - I * (I + 1) / 2;
        const int R = 16 * I + (lane >> 4) + 4 * reg, Cc = 16 * J + (lane & 15);
        int off = -1, dgA = -1, dgRC = 0;
        if (R < n6 && Cc < n6 && R >= Cc) {
            const int fa = Cc / 6, c = Cc - 6 * fa, fb = R / 6, r = R - 6 * fb, bo = fb - fa;
            if (bo > 0 || c <= r) { off = tvis_col(fa, N) + bo * 36 + r * 6 + c; if (bo == 0) { dgA = fa; dgRC = r * 6 + c; } }
        } else if (R == n6 && Cc < n6) off = -2 - Cc;
        if (off == -1) continue;
        // (eight group partials in flight; masked adds in group order)
        double s = 0.0;
        for (int g0 = 0; g0 < Gr; g0 += 8) {
            double p8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) p8[u] = r1[(size_t)(g0 + u < Gr ? g0 + u : Gr - 1) * (ntiles * 256) + e];
#pragma unroll
            for (int u = 0; u < 8; u++) if (g0 + u < Gr) s = (g0 + u == 0) ? p8[u] : s + p8[u];
        }
        if (off >= 0) {
            const double cur = (from_partials && dgA >= 0) ? diag_entry(dgA, dgRC) : out[off];
            out[off] = cur - s;
        } else out[tail + 12 * N + (-2 - off)] = -s;
    }
}
