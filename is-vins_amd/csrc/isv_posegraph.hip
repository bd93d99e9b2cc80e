// isv_posegraph.hip -- PoseGraph::optimizeCS on the MI355X (reference src/pose_graph/pose_graph.cpp:234-428), the consumer
// of the CombinedFactors the sliding-window backend emits (C ABI: include/isvins_posegraph.h).
//
// The problem (Ceres 2.0.0 in the reference: SPARSE_NORMAL_CHOLESKY, LEVENBERG_MARQUARDT, <= 10 iterations): the poses
// of the keyframes first_looped_index .. cur_index; RelativePoseFactor chain + RollPitchFactor per keyframe + loop-closure
// RelativePoseFactors under HuberLoss(0.1); then ceres::Covariance of every pose block.
//
// Device design.  The normal equations are BLOCK SPARSE: a block-tridiagonal chain plus one long row per loop closure.
// With the keyframes in their natural order a Cholesky factor fills only inside the row ENVELOPE (row r: columns
// start[r] .. r, start[r] = its earliest neighbour), so the matrices live in skyline storage of 6x6 blocks: K chain
// blocks + sum over loop edges of (later - earlier).  ONE WAVEFRONT per pose graph runs the whole optimisation -- the
// factorisation is a recurrence along the chain, there is nothing for a second wavefront to do -- and a batch of graphs
// (one per sequence) fills the GPU: grid = graphs.  Inside the wavefront: lane per residual block for the
// linearisation (RelativePoseFactor / RollPitchFactor::Evaluate + Corrector), lane per pose for the assembly (owner
// computes, fixed adjacency order: bitwise reproducible), 36 lanes per 6x6 block product in the factorisation, the
// triangular solves and the selected inversion.  Marginal covariances = the diagonal blocks of H^-1 by the Takahashi
// recurrence on the same envelope (never a dense inverse).
// This is latency-bound sparse work (dependent 6x6 block recurrences through L2); it is not reshaped into dense GEMMs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/isvins_posegraph.h"
#include "isv_device_math.h"
#include "isv_prior_factor.h"

#define PG_WSYNC() ISV_WSYNC()
// LDS-only ordering inside the wavefront (the 6x6 tiles T0 / T1 / T2): the LDS executes one wavefront's accesses in issue order,
// so all that is needed is that the compiler neither reorders them nor keeps tile values in registers.  Unlike a release
// fence this does NOT wait for outstanding global loads and stores (vmcnt): with PG_WSYNC every step of the factorisation
// paid the latency of its own L-block store and of the next row's prefetch.
#define PG_LSYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
// make this wavefront's global stores visible to its own later loads (other lanes read what a lane wrote)
#define PG_GSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

struct PgEdge {                  // one residual block
    int32_t kind;                // 0 roll/pitch (a), 1 relative pose (a -> b), 2 loop closure (a = matched keyframe, b = the one that closed)
    int32_t a, b;                // local pose indices
    int32_t fa, fb;              // their free (non-constant) indices, -1 if constant
    int32_t dim, robust, _pad;
    double meas_t[3], meas_R[9], sqrt_info[36];
};

struct PgGraph {                 // offsets of one pose graph into the handle's pools
    int32_t P1, nf, ne, nblk;    // poses, free poses, residual blocks, skyline blocks
    int32_t pose0, free0, edge0, blk0, adj0, col0, vec0, _pad;
    int32_t max_iter, _pad2;
    double huber;
};

struct PgDev {
    PgGraph *graphs;
    double *pose, *cand;         // [poses][7]
    int32_t *free_of;            // [poses] free index of a local pose or -1
    PgEdge *edges;
    double *eres, *ejac;         // [edges][6], [edges][72]: corrected residual, Jacobians (block a | block b), Jacobi-scaled
    double *Bt;                  // [free poses][36]: B_c = L_cc^-1 L(c, c-1) of every chain row c (factor(): a loop row's step through column c)
    double *red;                 // [edges][6]: per-item terms of the sums a multi-wavefront launch hands to wavefront 0 (k_pgo<4>)
    int32_t *adj_ptr, *adj;      // per free pose: the residual blocks touching it, (edge << 1) | side
    int32_t *start, *rowptr;     // skyline: first column of row r, block offset of row r   (per free pose; rowptr has nf + 1)
    int32_t *colptr, *colrows;   // column pattern: rows i > j with start[i] <= j, ascending
    double *H, *L, *Z;           // [blocks][36]
    double *scale, *diag, *grad, *step, *ysol;   // [6 nf] each
    double *cov;                 // [poses][36] tangent-space marginal covariance (zero for constant poses)
    isv_pgo_result_t *res;
    int32_t idx_lds_rows, idx_lds_cols;   // capacity of the LDS copies of start / rowptr / colptr and of colrows (0: read them from global memory)
    int32_t g0, _pad;                     // first graph of this launch (a batch call launches its graphs in chunks, round 4)
};

// ---- 6x6 helpers: lane e < 36 owns element (a, b) = (e / 6, e % 6) -------------------------------------------------
DEV double readlane_d(double v, int lane) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
DEV double pg_rsqrt(double x) {                       // 1/sqrt(x) to ~1 ulp: hardware estimate + two Newton steps (no f64 sqrt / divide sequences on the serial path)
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
// D (6x6 SPD, row-major in LDS tile T, lower part valid) -> T = inverse of its Cholesky factor (lower); returns false if not SPD
DEV bool pg_chol_inv6(double *T, int lane) {
    double row[6], dinv[6], x[6];
#pragma unroll
    for (int k = 0; k < 6; k++) row[k] = lane < 6 ? T[lane * 6 + k] : 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double s = row[j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= row[k] * readlane_d(row[k], j);
        const double sj = readlane_d(s, j);
        if (!(sj > 0.0)) bad = true;
        dinv[j] = pg_rsqrt(sj);
        row[j] = (lane == j) ? sj * dinv[j] : s * dinv[j];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {                      // lane c solves L x = e_c
        double s = (lane == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) s -= readlane_d(row[k], i) * x[k];
        x[i] = s * dinv[i];
    }
    PG_LSYNC();
    if (lane < 6) {
#pragma unroll
        for (int k = 0; k < 6; k++) T[k * 6 + lane] = x[k];      // x[k] = Linv[k][lane], zero for k < lane
    }
    PG_LSYNC();
    return !bad;
}
// deterministic wave sum (fixed butterfly)
DEV double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
DEV double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// RelativePoseFactor::Evaluate (include/factor/relative_pose_factor.h:27-70) / RollPitchFactor::Evaluate
// (rollpitch_factor.h:26-57) of one residual block at `pose`, HuberLoss corrector applied; returns rho(s) / 2
// Every global input is read into registers before the first store and the outputs are stored at the very end: the
// compiler must keep loads behind earlier stores that may alias them, and a store -> dependent load round trip per matrix
// entry is what the first version of this kernel spent most of its time on.
DEV double pg_edge_eval(const PgEdge &E, const double *pose, double huber, double *__restrict__ r_out, double *__restrict__ Ja, double *__restrict__ Jb, bool jac) {
    double raw[6], rJa[36], rJb[36], Sq[36];
    const double *pa = pose + 7 * E.a, *pb = pose + 7 * E.b;
    const int dim = E.dim, kind = E.kind, robust = E.robust;
    // (fixed 6 x 6 shapes, zero-padded for the 2-row roll/pitch factor: constant loop bounds keep the arrays in registers,
    //  and the padding adds exact zeros to the sums)
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int k = 0; k < 6; k++) Sq[a * 6 + k] = (a < dim && k < dim) ? E.sqrt_info[a * dim + k] : 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) raw[k] = 0.0;
    if (kind == 0) {
        Quat Rq = q_normalized(q_from_pose(pa)), Rm = q_from_R(E.meas_R);
        double nZ[3] = {0, 0, -1.0}, v[3];
        q_rot(so3_mul(Rm, q_conj(Rq)), nZ, v);
        raw[0] = v[0]; raw[1] = v[1];
        if (jac) {
            double S[9], Rmm[9], Bm[9];
            skew3(v, S); q_to_R(Rm, Rmm); m3_mul(S, Rmm, Bm);
            for (int k = 0; k < 36; k++) rJa[k] = 0;
            for (int a = 0; a < 2; a++) for (int b = 0; b < 3; b++) rJa[a * 6 + 3 + b] = Bm[a * 3 + b];
        }
    } else {
        relpose_jac(E.meas_t, E.meas_R, pa, pb, raw, rJa, rJb);
    }
    double r[6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += Sq[a * 6 + k] * raw[k];
        r[a] = s;
    }
    double sq = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) sq += r[a] * r[a];
    double rho = sq, sc = 1.0;
    if (robust) {                                    // HuberLoss(a): rho = s (s <= a^2), 2 a sqrt(s) - a^2 beyond; Corrector scales by sqrt(rho')
        const double b2 = huber * huber;
        if (sq > b2) { const double rr = sqrt(sq); rho = 2.0 * huber * rr - b2; sc = sqrt(fmax(2.2250738585072014e-308, huber / rr)); }
    }
    if (jac) {
        double oa[36], ob[36];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) s += Sq[a * 6 + k] * rJa[k * 6 + c];
                oa[a * 6 + c] = s * sc;
            }
        if (kind != 0) {
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) s += Sq[a * 6 + k] * rJb[k * 6 + c];
                    ob[a * 6 + c] = s * sc;
                }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) if (a < dim) r_out[a] = r[a] * sc;
#pragma unroll
        for (int q = 0; q < 36; q++) if (q < dim * 6) Ja[q] = oa[q];
        if (kind != 0) {
#pragma unroll
            for (int q = 0; q < 36; q++) Jb[q] = ob[q];
        }
    }
    return 0.5 * rho;
}

// NW wavefronts per pose graph (round 5).  NW = 1: the batch form, a graph per wavefront.  NW = 4: a call with few graphs -- what
// PoseGraph::optimizeCS is in the reference: ONE graph, when a loop closes -- leaves the GPU idle around one wavefront per graph, so
// the loops that are lane-per-residual-block or lane-per-pose (evaluation + linearisation, column norms, scaling, gradient, assembly,
// candidate, the element-wise passes, the covariance read-out) stride over four wavefronts; the block recurrences (factor, solve,
// the selected inverse) stay on wavefront 0: row r needs row r - 1.  SAME BITS as NW = 1: the owner-computes loops have no
// cross-lane sum, and every sum over items is taken by wavefront 0 over the stored per-item terms in the order NW = 1 adds them.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_pgo(PgDev dv) {
    __shared__ double T0[36], T1[36], T2[36];
    __shared__ double bcast_[2 + NW];
    constexpr int NT = 64 * NW;
    const int gid = blockIdx.x + dv.g0;
    const PgGraph G = dv.graphs[gid];
    const int tid = threadIdx.x, lane = tid & 63, wv = NW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0, nf = G.nf, n = 6 * nf;
    // block-wide ordering of global + LDS traffic (NW = 1: the wavefront's own)
#define PG_BSYNC() do { if constexpr (NW > 1) __syncthreads(); else PG_GSYNC(); } while (0)
    // a value wavefront 0 holds -> every wavefront
    auto bc = [&](double v) -> double {
        if constexpr (NW == 1) return v;
        if (tid == 0) bcast_[0] = v;
        __syncthreads();
        const double r = bcast_[0];
        __syncthreads();
        return r;
    };
    auto block_max = [&](double m) -> double {          // (a maximum does not depend on the order)
        m = wave_max(m);
        if constexpr (NW == 1) return m;
        if (lane == 0) bcast_[2 + wv] = m;
        __syncthreads();
        double r = bcast_[2];
        for (int k = 1; k < NW; k++) r = fmax(r, bcast_[2 + k]);
        __syncthreads();
        return r;
    };
    double *red = dv.red + (size_t)G.edge0 * 6;
    double *Bt = dv.Bt + (size_t)G.vec0 * 6;
    double *pose = dv.pose + (size_t)G.pose0 * 7, *cand = dv.cand + (size_t)G.pose0 * 7;
    const int32_t *free_of = dv.free_of + G.pose0;
    PgEdge *edges = dv.edges + G.edge0;
    double *eres = dv.eres + (size_t)G.edge0 * 6, *ejac = dv.ejac + (size_t)G.edge0 * 72;
    const int32_t *adj_ptr = dv.adj_ptr + G.free0 + gid, *adj = dv.adj + G.adj0;
    const int32_t *start = dv.start + G.free0, *rowptr = dv.rowptr + G.free0 + gid;
    const int32_t *colptr = dv.colptr + G.free0 + gid, *colrows = dv.colrows + G.col0;
    double *H = dv.H + (size_t)G.blk0 * 36, *L = dv.L + (size_t)G.blk0 * 36, *Z = dv.Z + (size_t)G.blk0 * 36;
    double *scale = dv.scale + G.vec0, *diag = dv.diag + G.vec0, *grad = dv.grad + G.vec0, *step = dv.step + G.vec0, *ysol = dv.ysol + G.vec0;
    isv_pgo_result_t &res = dv.res[gid];
    const int e = lane, ea = e / 6, eb = e - 6 * ea;    // my element of a 6x6 block (lanes 0..35)
    // the envelope's index arrays in LDS when they fit (every block address of the factorisation / substitution recurrences
    // starts from them: one LDS read instead of a dependent global load per step)
    extern __shared__ int32_t dyn_idx[];
    const bool idx_lds = dv.idx_lds_rows > 0;          // (sized by the host for the largest graph of the batch)
    const int IR = dv.idx_lds_rows;
    if (idx_lds) {
        const int nc = colptr[nf];
        for (int q = tid; q < nf; q += NT) { dyn_idx[q] = start[q]; dyn_idx[IR + q] = rowptr[q]; }
        for (int q = tid; q <= nf; q += NT) dyn_idx[2 * IR + q] = colptr[q];
        for (int q = tid; q < nc; q += NT) dyn_idx[3 * IR + 1 + q] = colrows[q];
        if constexpr (NW > 1) __syncthreads(); else PG_WSYNC();
    }
    // (explicit LDS reads: a pointer that may be LDS or global becomes a FLAT access, which waits for every outstanding
    //  global load and store as well)
    auto ST = [&](int r) -> int { return idx_lds ? dyn_idx[r] : start[r]; };
    auto RP = [&](int r) -> int { return idx_lds ? dyn_idx[IR + r] : rowptr[r]; };
    auto CP = [&](int r) -> int { return idx_lds ? dyn_idx[2 * IR + r] : colptr[r]; };
    auto CR = [&](int q) -> int { return idx_lds ? dyn_idx[3 * IR + 1 + q] : colrows[q]; };
    auto BLK = [&](double *M, int r, int c) -> double * { return M + (size_t)(RP(r) + (c - ST(r))) * 36; };

    // ---- evaluation of every residual block at x (and linearisation) ------------------------------------------------
    auto evaluate = [&](const double *x, bool jac) -> double {
        double c = 0;
        if constexpr (NW == 1) {
            for (int q = lane; q < G.ne; q += 64) c += pg_edge_eval(edges[q], x, G.huber, eres + 6 * q, ejac + 72 * q, ejac + 72 * q + 36, jac);
            PG_GSYNC();
            return wave_sum(c);
        } else {
            for (int q = tid; q < G.ne; q += NT) red[q] = pg_edge_eval(edges[q], x, G.huber, eres + 6 * q, ejac + 72 * q, ejac + 72 * q + 36, jac);
            __syncthreads();
            if (wv == 0) { for (int q = lane; q < G.ne; q += 64) c += red[q]; c = wave_sum(c); }
            return bc(c);
        }
    };
    // squared column norms of the (current) Jacobian, owner computes: lane per free pose
    auto colnorm2 = [&](double *out) {
        for (int f = tid; f < nf; f += NT) {
            double s[6] = {0, 0, 0, 0, 0, 0};
            for (int q = adj_ptr[f]; q < adj_ptr[f + 1]; q++) {
                const int ed = adj[q] >> 1, side = adj[q] & 1, dim = edges[ed].dim;
                const double *J = ejac + 72 * ed + 36 * side;
                for (int a = 0; a < dim; a++) for (int c = 0; c < 6; c++) s[c] += J[a * 6 + c] * J[a * 6 + c];
            }
            for (int c = 0; c < 6; c++) out[6 * f + c] = s[c];
        }
        PG_BSYNC();
    };
    auto scale_jac = [&]() {                           // J <- J diag(scale), lane per residual block
        for (int q = tid; q < G.ne; q += NT) {
            const PgEdge &E = edges[q];
            for (int side = 0; side < (E.kind == 0 ? 1 : 2); side++) {
                const int f = side ? E.fb : E.fa;
                if (f < 0) continue;
                double *J = ejac + 72 * q + 36 * side;
                double sc6[6], Jl[36];
                const int nq = E.dim * 6;
#pragma unroll
                for (int c = 0; c < 6; c++) sc6[c] = scale[6 * f + c];
#pragma unroll
                for (int k = 0; k < 36; k++) Jl[k] = k < nq ? J[k] : 0.0;
#pragma unroll
                for (int k = 0; k < 36; k++) if (k < nq) J[k] = Jl[k] * sc6[k % 6];
            }
        }
        PG_BSYNC();
    };
    // gradient g = J^T r (scaled Jacobian), lane per free pose; returns nothing
    auto gradient = [&]() {
        for (int f = tid; f < nf; f += NT) {
            double s[6] = {0, 0, 0, 0, 0, 0};
            for (int q = adj_ptr[f]; q < adj_ptr[f + 1]; q++) {
                const int ed = adj[q] >> 1, side = adj[q] & 1, dim = edges[ed].dim;
                const double *J = ejac + 72 * ed + 36 * side, *r = eres + 6 * ed;
                for (int a = 0; a < dim; a++) for (int c = 0; c < 6; c++) s[c] += J[a * 6 + c] * r[a];
            }
            for (int c = 0; c < 6; c++) grad[6 * f + c] = s[c];
        }
        PG_BSYNC();
    };
    // gradient_max_norm = |x - Plus(x, -g_unscaled)|_inf over the free blocks
    auto gmax_of = [&]() -> double {
        double m = 0;
        for (int k = tid; k < G.P1; k += NT) {
            const int f = free_of[k];
            if (f < 0) continue;
            double ng[6], xp[7];
            for (int c = 0; c < 6; c++) ng[c] = -grad[6 * f + c] / scale[6 * f + c];
            pose_plus(pose + 7 * k, ng, xp);
            for (int c = 0; c < 7; c++) m = fmax(m, fabs(pose[7 * k + c] - xp[c]));
        }
        return block_max(m);
    };
    auto xnorm_of = [&](const double *x) -> double {
        double s = 0;
        if (wv == 0) { for (int k = lane; k < G.P1; k += 64) if (free_of[k] >= 0) for (int c = 0; c < 7; c++) s += x[7 * k + c] * x[7 * k + c]; s = wave_sum(s); }
        return sqrt(bc(s));
    };
    // H = J^T J on the envelope, lane per free pose (row): its diagonal block and the blocks towards EARLIER neighbours
    auto assemble = [&]() {
        for (int q = tid; q < G.nblk * 36; q += NT) H[q] = 0.0;
        PG_BSYNC();
        for (int f = tid; f < nf; f += NT) {
            for (int q = adj_ptr[f]; q < adj_ptr[f + 1]; q++) {
                const int ed = adj[q] >> 1, side = adj[q] & 1;
                const PgEdge &E = edges[ed];
                const double *J = ejac + 72 * ed + 36 * side;
                const int dim = E.dim, kind = E.kind, other = kind != 0 ? (side ? E.fa : E.fb) : -1;
                double Jl[36], blk[36], old[36];
#pragma unroll
                for (int k = 0; k < 36; k++) Jl[k] = k < dim * 6 ? J[k] : 0.0;     // (rows beyond dim: exact zeros in the sums)
                double *D = BLK(H, f, f);
#pragma unroll
                for (int k = 0; k < 36; k++) old[k] = D[k];
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = 0; b <= a; b++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 6; k++) s += Jl[k * 6 + a] * Jl[k * 6 + b];
                        blk[a * 6 + b] = s; blk[b * 6 + a] = s;
                    }
#pragma unroll
                for (int k = 0; k < 36; k++) D[k] = old[k] + blk[k];
                if (other >= 0 && other < f) {                 // the later endpoint owns the off-diagonal block (f, other)
                    const double *Jo = ejac + 72 * ed + 36 * (1 - side);
                    double Jol[36];
#pragma unroll
                    for (int k = 0; k < 36; k++) Jol[k] = Jo[k];
                    double *O = BLK(H, f, other);
#pragma unroll
                    for (int k = 0; k < 36; k++) old[k] = O[k];
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int b = 0; b < 6; b++) {
                            double s = 0;
#pragma unroll
                            for (int k = 0; k < 6; k++) s += Jl[k * 6 + a] * Jol[k * 6 + b];
                            blk[a * 6 + b] = s;
                        }
#pragma unroll
                    for (int k = 0; k < 36; k++) O[k] = old[k] + blk[k];
                }
            }
        }
        PG_BSYNC();
    };
    // L L^T = H + diag(damp) on the envelope; diagonal slots hold L_rr^-1.  Returns false when a pivot is not positive.
    // The recurrence runs through LDS: a CHAIN row (its envelope starts at r - 1) needs L_{r-1,r-1}^-1 (still in T2 from the
    // previous row) and its own new block (T1), nothing from global memory that this wavefront wrote since the last fence;
    // its two blocks of H are loaded one row ahead.  Only a LOOP row (longer envelope) reads earlier rows' blocks of L back
    // from global memory and pays for a fence: one when it starts, one before its diagonal block, and one per step whose
    // column is itself a loop row.
    auto factor = [&](const double *damp) -> bool {
        bool ok = true;                                // (every block of L is written before it is read: A comes from H)
        double nx_rc = 0, nx_rr = 0;                   // H(r, r-1)[e], H(r, r)[e] (+ damping) of the NEXT row when it is a chain row
        bool have_nx = false;
        for (int r = 0; r < nf && ok; r++) {
            const int s0 = ST(r);
            const bool loop_row = s0 < r - 1;
            const double a_rc = nx_rc, a_rr = nx_rr;
            const bool pre = have_nx;
            have_nx = false;
            if (r + 1 < nf && ST(r + 1) == r) {        // prefetch the next chain row's blocks
                if (e < 36) { nx_rc = BLK(H, r + 1, r)[e]; nx_rr = BLK(H, r + 1, r + 1)[e] + (ea == eb ? damp[6 * (r + 1) + ea] : 0.0); }
                have_nx = true;
            }
            if (loop_row) PG_GSYNC();                  // blocks of earlier rows written since the last fence
            // one column of row r the general way: S = A(r,c) - sum_{m = max(start[r], start[c])}^{c-1} L(r,m) L(c,m)^T ;  L(r,c) = S L_cc^-T
            auto general_step = [&](int c) {
                const int sc = ST(c), m0 = s0 > sc ? s0 : sc;
                if (m0 < c - 1) PG_GSYNC();            // several blocks of THIS row are read back (column c is a loop row)
                double acc = 0;
                if (e < 36) {
                    acc = (pre && c == r - 1) ? a_rc : BLK(H, r, c)[e];
                    for (int m = m0; m < c - 1; m++) {
                        const double *X = BLK(L, r, m), *Y = BLK(L, c, m);
                        for (int k = 0; k < 6; k++) acc -= X[ea * 6 + k] * Y[eb * 6 + k];
                    }
                    if (m0 < c) {                      // m = c - 1: the block L(r, c-1) of the previous step is in T1 (LDS)
                        const double *Y = BLK(L, c, c - 1);
                        for (int k = 0; k < 6; k++) acc -= T1[ea * 6 + k] * Y[eb * 6 + k];
                    }
                }
                PG_LSYNC();
                if (e < 36) T0[e] = acc;
                PG_LSYNC();
                double x = 0;
                if (e < 36) {
                    if (c == r - 1) { for (int k = 0; k <= eb; k++) x += T0[ea * 6 + k] * T2[eb * 6 + k]; }       // L_cc^-1 of the previous row (LDS)
                    else { const double *Li = BLK(L, c, c); for (int k = 0; k <= eb; k++) x += T0[ea * 6 + k] * Li[eb * 6 + k]; }
                }
                PG_LSYNC();
                if (e < 36) { T1[e] = x; BLK(L, r, c)[e] = x; }
                PG_LSYNC();
            };
            if (!loop_row) { for (int c = s0; c < r; c++) general_step(c); }
            else {
                // (round 5) A loop closure's row walks its envelope column by column, every step waiting for the one before: ~900 of the
                // dependent block steps of a factorisation of 200 keyframes / 5 loops, and each took one memory latency (its operands were
                // loaded ONE step ahead: 0.45 us per step whatever the arithmetic).  An interior column c is a chain row this row has no
                // factor with -- H(r,c) = 0 --, so L(r,c) = -L(r,c-1) L(c,c-1)^T L_cc^-T = -L(r,c-1) B_c^T with B_c left behind by row c:
                // ONE 6 x 6 product and two LDS hand-overs.  H(r,c) and B_c are loaded PD steps ahead into a register ring (the
                // envelope is contiguous in both); any other column takes general_step (it loads what it needs itself).
                constexpr int PD = 8;
                double rA[PD], rB[PD][6];
                // (the row's blocks of H and L and the chain rows' B are contiguous in c: one base address each, no index look-ups per step)
                const int el = e < 36 ? e : 0;
                const double *const Hr = BLK(H, r, s0) + el, *const Br = Bt + (size_t)s0 * 36 + (el % 6) * 6;
                double *const Lr = BLK(L, r, s0) + el;
                auto prefetch = [&](double &pa, double *pb, int c) {
                    const int dc = (c < r ? c : r - 1) - s0;                            // (past the row: a valid address, never used)
                    pa = Hr[dc * 36];
                    for (int k = 0; k < 6; k++) pb[k] = Br[dc * 36 + k];
                };
                general_step(s0);
#pragma unroll
                for (int u = 0; u < PD; u++) prefetch(rA[u], rB[u], s0 + 1 + u);
                for (int cb = s0 + 1; cb < r; cb += PD) {
#pragma unroll
                    for (int u = 0; u < PD; u++) {
                        const int c = cb + u;
                        if (c < r) {
                            const double cA = rA[u];
                            double cB[6];
                            for (int k = 0; k < 6; k++) cB[k] = rB[u][k];
                            prefetch(rA[u], rB[u], c + PD);
                            if (ST(c) == c - 1 && __all(e >= 36 || cA == 0.0)) {
                                double x = 0;
                                if (e < 36) { for (int k = 0; k < 6; k++) x -= T1[ea * 6 + k] * cB[k]; }
                                PG_LSYNC();
                                if (e < 36) { T1[e] = x; Lr[(c - s0) * 36] = x; }
                                PG_LSYNC();
                            } else general_step(c);
                        }
                    }
                }
            }
            // diagonal: D = A(r,r) + damp - sum_m L(r,m) L(r,m)^T
            if (loop_row) PG_GSYNC();
            double acc = 0;
            if (e < 36) {
                acc = pre ? a_rr : BLK(H, r, r)[e] + (ea == eb ? damp[6 * r + ea] : 0.0);
                for (int m = s0; m < r - 1; m++) { const double *X = BLK(L, r, m); for (int k = 0; k < 6; k++) acc -= X[ea * 6 + k] * X[eb * 6 + k]; }
                if (s0 < r) { for (int k = 0; k < 6; k++) acc -= T1[ea * 6 + k] * T1[eb * 6 + k]; }                // m = r - 1
            }
            PG_LSYNC();                                // (T2 = L_{r-1,r-1}^-1 has been read by every lane)
            if (e < 36) T2[e] = acc;
            PG_LSYNC();
            ok = pg_chol_inv6(T2, lane);
            if (e < 36) {
                BLK(L, r, r)[e] = T2[e];
                // B_r = L_rr^-1 L(r, r-1) for the loop rows that will pass through column r (zero where row r has no earlier block)
                double b = 0;
                if (s0 == r - 1) { for (int k = 0; k <= ea; k++) b += T2[ea * 6 + k] * T1[k * 6 + eb]; }
                Bt[(size_t)r * 36 + e] = b;
            }
        }
        PG_GSYNC();
        return ok;
    };
    // x = (L L^T)^-1 b  -> xs ; lanes 0..5 own the components of a block.  The previous block of the recurrence stays in
    // registers (v_readlane); only loop rows exchange through global memory (fences as in factor()).
    // (round 5) The blocks of L a row needs -- its diagonal block and the one towards its chain neighbour -- and its right-hand side
    // are loaded PS rows AHEAD into a register ring: with the loads one row ahead (forward) or not ahead at all (backward) every one
    // of the 2 nf steps waited for a memory latency, 0.74 us per step for a few dozen FMAs.  The backward pass re-reads the ring's
    // right-hand sides after a loop row has updated them.
    auto solve = [&](const double *b, double *xs) {
        const int ln = lane < 6 ? lane : 0;
        constexpr int PS = 8;
        double prev = 0;                               // y_{r-1} (forward) / x_{r+1} (backward), component = lane
        {
            double fX[PS][6], fLi[PS][6], fb[PS];      // row ln of L(r, r-1), row ln of L_rr^-1, b_r of the rows r .. r + PS - 1
            auto fpf = [&](double *X6, double *Li6, double &bb, int rr) {
                const int q = rr < nf ? rr : nf - 1;
                const double *Li = BLK(L, q, q), *X = ST(q) < q ? BLK(L, q, q - 1) : Li;          // (no earlier block: a valid address, never used)
                for (int k = 0; k < 6; k++) { X6[k] = X[ln * 6 + k]; Li6[k] = Li[ln * 6 + k]; }
                bb = b[6 * q + ln];
            };
#pragma unroll
            for (int u = 0; u < PS; u++) fpf(fX[u], fLi[u], fb[u], u);
            for (int r0 = 0; r0 < nf; r0 += PS) {      // forward: y_r = L_rr^-1 (b_r - sum_{m<r} L(r,m) y_m)
#pragma unroll
                for (int u = 0; u < PS; u++) {
                    const int r = r0 + u;
                    if (r < nf) {
                        const int s0 = ST(r);
                        double cX[6], cLi[6];
                        const double cb = fb[u];
                        for (int k = 0; k < 6; k++) { cX[k] = fX[u][k]; cLi[k] = fLi[u][k]; }
                        fpf(fX[u], fLi[u], fb[u], r + PS);
                        if (s0 < r - 1) PG_GSYNC();
                        double pv[6];
                        for (int k = 0; k < 6; k++) pv[k] = readlane_d(prev, k);
                        double v = cb;
                        if (s0 == r - 1) { for (int k = 0; k < 6; k++) v -= cX[k] * pv[k]; }
                        else {
                            for (int m = s0; m < r; m++) {
                                const double *X = BLK(L, r, m);
                                if (m == r - 1) { for (int k = 0; k < 6; k++) v -= X[ln * 6 + k] * pv[k]; }
                                else { for (int k = 0; k < 6; k++) v -= X[ln * 6 + k] * ysol[6 * m + k]; }
                            }
                        }
                        double y = 0;
                        for (int k = 0; k < 6; k++) { const double vk = readlane_d(v, k); if (k <= ln) y += cLi[k] * vk; }
                        if (lane < 6) ysol[6 * r + lane] = y;
                        prev = y;
                    }
                }
            }
        }
        PG_GSYNC();
        // backward: x_r = L_rr^-T (y_r - sum_{i in colpat(r)} L(i,r)^T x_i).  A loop row i subtracts its L(i,c)^T x_i from y_c of
        // every column c it covers as soon as x_i is known (lanes over the span); what is left for a row is the term of row
        // r + 1 when that is a chain row, from registers.
        prev = 0;
        {
            double gLi[PS][6], gX[PS][6], gy[PS];      // column ln of L_rr^-1, column ln of L(r+1, r), y_r of the rows r .. r - PS + 1
            auto bpf = [&](double *Li6, double *X6, double &yy, int rr) {
                const int q = rr >= 0 ? rr : 0;
                const double *Li = BLK(L, q, q), *X = (q + 1 < nf && ST(q + 1) == q) ? BLK(L, q + 1, q) : Li;      // (row q + 1 not a chain row: never used)
                for (int k = 0; k < 6; k++) { Li6[k] = Li[k * 6 + ln]; X6[k] = X[k * 6 + ln]; }
                yy = ysol[6 * q + ln];
            };
#pragma unroll
            for (int u = 0; u < PS; u++) bpf(gLi[u], gX[u], gy[u], nf - 1 - u);
            bool next_is_chain = false;                // row r + 1 exists and its envelope starts at r
            for (int r0 = nf - 1; r0 >= 0; r0 -= PS) {
#pragma unroll
                for (int u = 0; u < PS; u++) {
                    const int r = r0 - u;
                    if (r >= 0) {
                        double cLi[6], cX[6];
                        double v = gy[u];
                        for (int k = 0; k < 6; k++) { cLi[k] = gLi[u][k]; cX[k] = gX[u][k]; }
                        bpf(gLi[u], gX[u], gy[u], r - PS);
                        double pv[6];
                        for (int k = 0; k < 6; k++) pv[k] = readlane_d(prev, k);
                        if (next_is_chain) { for (int k = 0; k < 6; k++) v -= cX[k] * pv[k]; }
                        double x = 0;
                        for (int k = 0; k < 6; k++) { const double vk = readlane_d(v, k); if (k >= ln) x += cLi[k] * vk; }
                        if (lane < 6) xs[6 * r + lane] = x;
                        prev = x;
                        const int s0 = ST(r);
                        next_is_chain = s0 == r - 1;
                        if (s0 < r - 1) {              // loop row: y_c -= L(r,c)^T x_r for c in [s0, r)
                            double xr[6];
                            for (int k = 0; k < 6; k++) xr[k] = readlane_d(x, k);
                            PG_GSYNC();
                            for (int q = lane; q < (r - s0) * 6; q += 64) {
                                const int c = s0 + q / 6, comp = q % 6;
                                const double *X = BLK(L, r, c);
                                double a = ysol[6 * c + comp];
                                for (int k = 0; k < 6; k++) a -= X[k * 6 + comp] * xr[k];
                                ysol[6 * c + comp] = a;
                            }
                            PG_GSYNC();
                            // the ring's right-hand sides were read before this update: slot j holds row r0 - j (j > u) or r0 - PS - j (already refilled)
#pragma unroll
                            for (int j = 0; j < PS; j++) { const int rowj = j <= u ? r0 - PS - j : r0 - j; gy[j] = ysol[6 * (rowj >= 0 ? rowj : 0) + ln]; }
                        }
                    }
                }
            }
        }
        PG_GSYNC();
    };

    // ================= TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy (Ceres 2.0.0 defaults) =============
    if (tid == 0) { res.status = ISV_OK; res.num_successful = 0; res.n_loop_edges = 0; res._pad = 0;     // (every byte of the record is defined: callers compare records)
                    for (int k = 0; k < ISV_MAX_TRACE; k++) { res.trace_cost[k] = 0; res.trace_accepted[k] = 0; } }
    double x_cost = evaluate(pose, true);
    int it = 0, term = ISV_TERM_RUNNING, nsucc = 0;
    if (nf > 0) {
        colnorm2(scale);
        for (int q = tid; q < n; q += NT) scale[q] = 1.0 / (1.0 + sqrt(scale[q]));
        PG_BSYNC();
        scale_jac();
        gradient();
        double gmax = gmax_of(), x_norm = xnorm_of(pose);
        double radius = 1e4, decrease = 2.0;
        bool reuse_diag = false;
        int invalid = 0;
        if (tid == 0) { res.initial_cost = x_cost; res.trace_cost[0] = x_cost; }
        for (;;) {
            if (it >= G.max_iter) { term = ISV_TERM_MAX_ITERATIONS; break; }
            if (gmax <= 1e-10) { term = ISV_TERM_GRADIENT_TOL; break; }
            if (radius <= 1e-32) { term = ISV_TERM_MIN_RADIUS; break; }
            it++;
            if (!reuse_diag) {                         // diagonal_ = clamp(squared column norms)
                colnorm2(diag);
                for (int q = tid; q < n; q += NT) diag[q] = fmin(fmax(diag[q], 1e-6), 1e32);
                PG_BSYNC();
            }
            reuse_diag = true;
            assemble();
            for (int q0 = tid; q0 < n; q0 += 4 * NT) {   // D^2 = diagonal / radius (step[] as scratch); four loads in flight per lane
                double v[4];
                for (int u = 0; u < 4; u++) v[u] = q0 + NT * u < n ? diag[q0 + NT * u] : 0.0;
                for (int u = 0; u < 4; u++) if (q0 + NT * u < n) step[q0 + NT * u] = v[u] / radius;
            }
            PG_BSYNC();
            bool ok = false;
            if (wv == 0) ok = factor(step);               // (the recurrences: wavefront 0)
            if (wv == 0 && ok) solve(grad, step);
            ok = bc(ok ? 1.0 : 0.0) != 0.0;               // (NW > 1: also the barrier that hands wavefront 0's stores to the others)
            if (ok) {
                double bad = 0;
                for (int q0 = tid; q0 < n; q0 += 4 * NT) {
                    double v[4];
                    for (int u = 0; u < 4; u++) v[u] = q0 + NT * u < n ? step[q0 + NT * u] : 0.0;
                    for (int u = 0; u < 4; u++) if (q0 + NT * u < n) { if (!(v[u] - v[u] == 0.0)) bad = 1; step[q0 + NT * u] = -v[u]; }
                }
                PG_BSYNC();
                if (block_max(bad) > 0) ok = false;
            }
            bool valid = false;
            double model_cost_change = 0;
            if (ok) {                                  // -(J step)^T (r + J step / 2)
                // (the term of a residual row is rounded before it is added -- no fused multiply-add across the two: what NW > 1 adds
                //  is the stored term)
                double mc = 0;
                for (int q = tid; q < G.ne; q += NT) {
                    const PgEdge &E = edges[q];
                    for (int a = 0; a < E.dim; a++) {
                        double m = 0;
                        if (E.fa >= 0) for (int c = 0; c < 6; c++) m += ejac[72 * q + a * 6 + c] * step[6 * E.fa + c];
                        if (E.kind != 0 && E.fb >= 0) for (int c = 0; c < 6; c++) m += ejac[72 * q + 36 + a * 6 + c] * step[6 * E.fb + c];
                        double term = m * (eres[6 * q + a] + m / 2.0);
                        asm volatile("" : "+v"(term));
                        if constexpr (NW == 1) mc += term; else red[6 * q + a] = term;
                    }
                }
                if constexpr (NW > 1) {
                    __syncthreads();
                    if (wv == 0) for (int q = lane; q < G.ne; q += 64) { const int dim = edges[q].dim; for (int a = 0; a < dim; a++) mc += red[6 * q + a]; }
                }
                model_cost_change = -bc(wave_sum(mc));
                valid = model_cost_change > 0.0;
            }
            if (!valid) {
                if (++invalid >= 5) { term = ok ? ISV_TERM_INVALID_STEPS : ISV_TERM_LINEAR_SOLVER; break; }
                radius /= decrease; decrease *= 2.0; reuse_diag = true;        // StepIsInvalid == StepRejected
                if (tid == 0 && it < ISV_MAX_TRACE) { res.trace_cost[it] = x_cost; res.trace_accepted[it] = 0; }
                continue;
            }
            invalid = 0;
            double dn = 0;
            for (int k = tid; k < G.P1; k += NT) {     // candidate = Plus(x, step * scale)
                const int f = free_of[k];
                if (f < 0) { double x0[7]; for (int c = 0; c < 7; c++) x0[c] = pose[7 * k + c]; for (int c = 0; c < 7; c++) cand[7 * k + c] = x0[c]; continue; }
                double dl[6], xp[7], x0[7];
                for (int c = 0; c < 6; c++) dl[c] = step[6 * f + c] * scale[6 * f + c];
                for (int c = 0; c < 7; c++) x0[c] = pose[7 * k + c];
                pose_plus(x0, dl, xp);
                if constexpr (NW == 1) { for (int c = 0; c < 7; c++) { const double df = x0[c] - xp[c]; dn = __builtin_fma(df, df, dn); } }
                for (int c = 0; c < 7; c++) cand[7 * k + c] = xp[c];
            }
            PG_BSYNC();
            if constexpr (NW > 1) {                    // |x - candidate|^2 in NW = 1's order, from the stored candidate
                if (wv == 0) for (int k = lane; k < G.P1; k += 64) {
                    if (free_of[k] < 0) continue;
                    for (int c = 0; c < 7; c++) { const double df = pose[7 * k + c] - cand[7 * k + c]; dn = __builtin_fma(df, df, dn); }
                }
            }
            const double step_norm = sqrt(bc(wave_sum(dn)));
            const double cand_cost = evaluate(cand, false);
            bool stop = false, accepted = false;
            if (step_norm <= 1e-8 * (x_norm + 1e-8)) { term = ISV_TERM_PARAMETER_TOL; stop = true; }
            else if (fabs(x_cost - cand_cost) <= 1e-6 * x_cost) { term = ISV_TERM_FUNCTION_TOL; stop = true; }
            if (stop) { if (tid == 0 && it < ISV_MAX_TRACE) { res.trace_cost[it] = x_cost; res.trace_accepted[it] = 0; } break; }
            const double rel = (x_cost - cand_cost) / model_cost_change;
            if (rel > 1e-3) {
                accepted = true;
                for (int q0 = tid; q0 < 7 * G.P1; q0 += 4 * NT) {
                    double v[4];
                    for (int u = 0; u < 4; u++) v[u] = q0 + NT * u < 7 * G.P1 ? cand[q0 + NT * u] : 0.0;
                    for (int u = 0; u < 4; u++) if (q0 + NT * u < 7 * G.P1) pose[q0 + NT * u] = v[u];
                }
                PG_BSYNC();
                x_norm = xnorm_of(pose);
                x_cost = evaluate(pose, true);
                scale_jac();
                gradient();
                gmax = gmax_of();
                radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3.0));
                radius = fmin(1e16, radius); decrease = 2.0; reuse_diag = false;
                nsucc++;
            } else { radius /= decrease; decrease *= 2.0; reuse_diag = true; }
            if (tid == 0 && it < ISV_MAX_TRACE) { res.trace_cost[it] = accepted ? x_cost : cand_cost; res.trace_accepted[it] = accepted ? 1 : 0; }
        }
    } else {
        term = ISV_TERM_GRADIENT_TOL;
        if (tid == 0) res.initial_cost = x_cost;
    }
    if (tid == 0) { res.iterations = it; res.termination = term; res.final_cost = x_cost; res.num_successful = nsucc; res.n_poses = G.P1; res.n_free = nf; }

    // ================= ceres::Covariance: diagonal blocks of (J^T J)^-1 at the solution, tangent space ==================
    double *cov = dv.cov + (size_t)G.pose0 * 36;
    for (int q = tid; q < G.P1 * 36; q += NT) cov[q] = 0.0;
    PG_BSYNC();
    if (nf == 0) return;
    // the stored Jacobian is the Jacobi-scaled one at x: H_s = S H S, so H^-1 = S H_s^-1 S
    assemble();
    for (int q = tid; q < n; q += NT) step[q] = 0.0;
    PG_BSYNC();
    bool fok = false;
    if (wv == 0) fok = factor(step);
    if (bc(fok ? 1.0 : 0.0) == 0.0) { if (tid == 0) res.status = ISV_ERR_NONFINITE; return; }
    // Takahashi recurrence on the envelope, columns from the last to the first:
    //   Z(i,j) = [delta_ij L_jj^-T - sum_{k in colpat(j)} Z(i,k) L(k,j)] L_jj^-1      for i in {j} U colpat(j)
    // (a recurrence over the columns: wavefront 0)
    // (round 5) A column that only its chain neighbour reaches down into -- colpat(j) = {j + 1} -- needs Z(j+1,j+1) (the diagonal block the
    // previous column ended with) and Z(j+1,j) (what its first step produces): both stay in LDS tiles (T1, T2), so such a column reads no
    // block of Z back from global memory and needs none of the two fences of the general column (each a memory round trip: 6.4 us per
    // column before); its blocks of L are loaded one column ahead.  Same operations in the same order as the general column: same bits.
    if (wv == 0) {
        bool dirty = false, have_pf = false;           // dirty: blocks of Z written since the last fence; have_pf: pLk / pLi / pLt hold column j's operands
        double pLk[6] = {0, 0, 0, 0, 0, 0}, pLi[6] = {0, 0, 0, 0, 0, 0}, pLt = 0;
        for (int j = nf - 1; j >= 0; j--) {
            const double *Li = BLK(L, j, j);
            const int c0 = CP(j), c1 = CP(j + 1);
            const bool fast = j + 1 < nf && c1 - c0 == 1 && CR(c0) == j + 1;      // (T1 = Z(j+1,j+1): every column leaves its diagonal block there)
            double lk[6], li[6], lt = pLt;
            for (int m = 0; m < 6; m++) { lk[m] = pLk[m]; li[m] = pLi[m]; }
            const bool pre = have_pf;
            have_pf = false;
            if (j >= 1 && ST(j) == j - 1 && e < 36) {                          // column j - 1's operands: L(j, j-1) (column eb), L_{j-1,j-1}^-1 (column eb; entry (eb, ea))
                const double *Ln = BLK(L, j, j - 1), *Lin = BLK(L, j - 1, j - 1);
                for (int m = 0; m < 6; m++) { pLk[m] = Ln[m * 6 + eb]; pLi[m] = Lin[m * 6 + eb]; }
                pLt = Lin[eb * 6 + ea];
            }
            if (j >= 1 && ST(j) == j - 1) have_pf = true;
            if (fast) {
                if (!pre && e < 36) {
                    const double *Lkj = BLK(L, j + 1, j);
                    for (int m = 0; m < 6; m++) { lk[m] = Lkj[m * 6 + eb]; li[m] = Li[m * 6 + eb]; }
                    lt = Li[eb * 6 + ea];
                }
                double acc = 0, x = 0;
                if (e < 36) { for (int m = 0; m < 6; m++) acc -= T1[ea * 6 + m] * lk[m]; }                 // Z(j+1,j) = -Z(j+1,j+1) L(j+1,j) L_jj^-1
                PG_LSYNC();
                if (e < 36) T0[e] = acc;
                PG_LSYNC();
                if (e < 36) { for (int m = eb; m < 6; m++) x += T0[ea * 6 + m] * li[m]; BLK(Z, j + 1, j)[e] = x; T2[e] = x; }
                PG_LSYNC();
                acc = lt; x = 0;                                                                             // Z(j,j) = (L_jj^-T - Z(j+1,j)^T L(j+1,j)) L_jj^-1
                if (e < 36) { for (int m = 0; m < 6; m++) acc -= T2[m * 6 + ea] * lk[m]; }
                if (e < 36) T0[e] = acc;
                PG_LSYNC();
                if (e < 36) { for (int m = eb; m < 6; m++) x += T0[ea * 6 + m] * li[m]; BLK(Z, j, j)[e] = x; T1[e] = x; }
                PG_LSYNC();
                dirty = true;
                continue;
            }
            if (dirty) { PG_GSYNC(); dirty = false; }
            for (int qi = c0; qi <= c1; qi++) {            // the rows below j first: Z(j,j) needs Z(k,j), k in colpat(j)
                const int i = qi < c1 ? CR(qi) : j;
                if (qi == c1) PG_GSYNC();
                double acc = 0;
                if (e < 36) {
                    if (i == j) acc = Li[eb * 6 + ea];                                   // L_jj^-T
                    for (int q = c0; q < c1; q++) {
                        const int k = CR(q);
                        const double *Lkj = BLK(L, k, j);
                        if (i >= k) { const double *Zik = BLK(Z, i, k); for (int m = 0; m < 6; m++) acc -= Zik[ea * 6 + m] * Lkj[m * 6 + eb]; }
                        else { const double *Zki = BLK(Z, k, i); for (int m = 0; m < 6; m++) acc -= Zki[m * 6 + ea] * Lkj[m * 6 + eb]; }
                    }
                }
                PG_LSYNC();
                if (e < 36) T0[e] = acc;
                PG_LSYNC();
                if (e < 36) {
                    double x = 0;
                    for (int m = eb; m < 6; m++) x += T0[ea * 6 + m] * Li[m * 6 + eb];   // times L_jj^-1 (lower)
                    BLK(Z, i, j)[e] = x;
                    if (i == j) T1[e] = x;
                }
            }
            PG_GSYNC();
        }
        if (dirty) PG_GSYNC();
    }
    if constexpr (NW > 1) __syncthreads();
    for (int k = tid; k < G.P1; k += NT) {
        const int f = free_of[k];
        if (f < 0) continue;
        const double *Zd = BLK(Z, f, f);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) cov[k * 36 + a * 6 + b] = scale[6 * f + a] * Zd[a * 6 + b] * scale[6 * f + b];
    }
}

// =====================================================================================================================
// host side
// the STRUCTURE of one graph slot as the last call analysed it (round 4: PoseGraph::optimizeCS re-optimises a graph that grew by
// at most one keyframe since the last loop closure; a batch that is optimised again -- the benchmark, a replay -- has the same
// topology slot by slot): parameter-block map, free indices, adjacency, skyline, column patterns and the edges' integer fields.
// Keyed by a hash of everything the structure depends on; the numbers (poses, measurements, sqrt_info) are refreshed every call.
struct PgEdgeTpl { int32_t kind, a, b, fa, fb, dim, robust, src; };     // src: the keyframe whose factor this residual block is
struct PgStructCache {
    uint64_t key = 0; bool valid = false;
    std::vector<uint64_t> raw;                    // the key tuple itself: a hash match alone would reuse another topology's structure on a collision
    std::vector<int> loc;
    int cur_pos = -1, n_loops = 0;
    int32_t P1 = 0, nf = 0, nblk = 0;
    std::vector<int32_t> free_of, adj_ptr, adj, start, rowptr, colptr, colrows;
    std::vector<PgEdgeTpl> edges;
};
struct isv_pgo {
    std::vector<PgStructCache> cache;             // [max_graphs]
    int64_t cache_hits = 0;
    isv_pgo_config_t cfg;
    std::string err;
    int device = 0, n_cus = 256;
    hipStream_t stream = nullptr;
    PgDev d{};
    std::vector<void *> allocs;
    size_t cap_pose = 0, cap_edge = 0, cap_blk = 0, cap_adj = 0, cap_col = 0;
    void *pin = nullptr; size_t pin_cap = 0;      // pinned staging of a batch call's arrays (grown on demand, kept)
    hipEvent_t ev[2] = {nullptr, nullptr};       // (unused since the chunked launches; kept for the destroy loop)
    // a batch call goes to the device in up to PG_CHUNKS chunks of graphs, each on a stream of its own: chunk c + 1 is assembled and
    // copied while chunk c is being solved, and chunk c is written back while chunk c + 1 still runs (round 4)
    static constexpr int PG_CHUNKS = 4;
    hipStream_t cstream[PG_CHUNKS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t kev[PG_CHUNKS][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};     // around every chunk's k_pgo
    double last_kernel_ms = -1;                   // first chunk's kernel start -> last chunk's kernel end of the last call
    double last_blocks = 0;                       // skyline blocks of the last batch (all graphs)
};

#define PCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return ISV_ERR_DEVICE; } } while (0)
template <typename T> static int pal(isv_pgo *h, T **p, size_t n) {
    void *q = nullptr;
    PCHK(h, hipMalloc(&q, (n ? n : 1) * sizeof(T)));
    h->allocs.push_back(q); *p = (T *)q;
    return ISV_OK;
}
#define PTRY(x) do { int rc_ = (x); if (rc_ != ISV_OK) return rc_; } while (0)

extern "C" const char *isv_pgo_last_error(const isv_pgo_t *h) { return h ? h->err.c_str() : "null handle"; }
// measurement: duration of k_pgo in the last optimize call (HIP events on the handle's stream) and the number of 6x6 skyline
// blocks its graphs held
extern "C" int64_t isv_pgo_structure_cache_hits(const isv_pgo_t *h) { return h ? h->cache_hits : 0; }
extern "C" int isv_pgo_last_kernel_ms(isv_pgo_t *h, double *ms, double *skyline_blocks) {
    if (!h || !ms || h->last_kernel_ms < 0) return ISV_ERR_INVALID_ARG;
    *ms = h->last_kernel_ms;
    if (skyline_blocks) *skyline_blocks = h->last_blocks;
    return ISV_OK;
}

extern "C" void isv_pgo_destroy(isv_pgo_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (void *p : h->allocs) (void)hipFree(p);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (auto &cs : h->cstream) if (cs) (void)hipStreamDestroy(cs);
    for (auto &ke : h->kev) for (auto &e : ke) if (e) (void)hipEventDestroy(e);
    delete h;
}

static int pgo_create_impl(isv_pgo *h) {
    int ndev = 0;
    PCHK(h, hipGetDeviceCount(&ndev));
    if (ndev <= 0) { h->err = "no HIP device"; return ISV_ERR_DEVICE; }
    PCHK(h, hipGetDevice(&h->device));
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cu > 0) h->n_cus = cu; else (void)hipGetLastError(); }
    PCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto &cs : h->cstream) PCHK(h, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    for (auto &ke : h->kev) for (auto &e : ke) PCHK(h, hipEventCreate(&e));
    const size_t G = h->cfg.max_graphs, K = h->cfg.max_keyframes;
    h->cap_pose = G * K; h->cap_edge = G * 3 * K; h->cap_blk = G * (2 * K + (size_t)h->cfg.max_loop_blocks);
    h->cap_adj = 2 * h->cap_edge; h->cap_col = h->cap_blk;
    PgDev &d = h->d;
    PTRY(pal(h, &d.graphs, G)); PTRY(pal(h, &d.pose, h->cap_pose * 7)); PTRY(pal(h, &d.cand, h->cap_pose * 7)); PTRY(pal(h, &d.free_of, h->cap_pose));
    PTRY(pal(h, &d.edges, h->cap_edge)); PTRY(pal(h, &d.eres, h->cap_edge * 6)); PTRY(pal(h, &d.ejac, h->cap_edge * 72)); PTRY(pal(h, &d.red, h->cap_edge * 6)); PTRY(pal(h, &d.Bt, h->cap_pose * 36));
    PTRY(pal(h, &d.adj_ptr, h->cap_pose + G)); PTRY(pal(h, &d.adj, h->cap_adj));
    PTRY(pal(h, &d.start, h->cap_pose)); PTRY(pal(h, &d.rowptr, h->cap_pose + G)); PTRY(pal(h, &d.colptr, h->cap_pose + G)); PTRY(pal(h, &d.colrows, h->cap_col));
    PTRY(pal(h, &d.H, h->cap_blk * 36)); PTRY(pal(h, &d.L, h->cap_blk * 36)); PTRY(pal(h, &d.Z, h->cap_blk * 36));
    PTRY(pal(h, &d.scale, h->cap_pose * 6)); PTRY(pal(h, &d.diag, h->cap_pose * 6)); PTRY(pal(h, &d.grad, h->cap_pose * 6));
    PTRY(pal(h, &d.step, h->cap_pose * 6)); PTRY(pal(h, &d.ysol, h->cap_pose * 6));
    PTRY(pal(h, &d.cov, h->cap_pose * 36)); PTRY(pal(h, &d.res, G));
    return ISV_OK;
}

extern "C" int isv_pgo_create(const isv_pgo_config_t *cfg, isv_pgo_t **out) {
    if (!cfg || !out) return ISV_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->max_keyframes < 2 || cfg->max_graphs < 1 || cfg->max_loop_blocks < 0 || cfg->max_iterations < 0 || cfg->max_iterations >= ISV_MAX_TRACE ||
        !(cfg->huber_delta > 0)) return ISV_ERR_INVALID_ARG;
    isv_pgo *h = new isv_pgo();
    h->cfg = *cfg;
    const int rc = pgo_create_impl(h);
    if (rc != ISV_OK) { fprintf(stderr, "isv_pgo_create: %s\n", h->err.c_str()); isv_pgo_destroy(h); return rc; }
    *out = h;
    return ISV_OK;
}

// ---- small host 3x3 / SO(3) arithmetic for the write-back (RelativePoseFactor::update, drift) ----------------------
namespace {
struct HQ { double w, x, y, z; };
void h_q2R(HQ q, double *R) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z, twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
HQ h_R2q(const double *m) {      // Eigen matrix -> quaternion
    HQ q; double t = m[0] + m[4] + m[8];
    if (t > 0) { t = std::sqrt(t + 1.0); q.w = 0.5 * t; t = 0.5 / t; q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t; return q; }
    int i = 0; if (m[4] > m[0]) i = 1; if (m[8] > m[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
    double v[3]; v[i] = 0.5 * t; t = 0.5 / t;
    q.w = (m[k * 3 + j] - m[j * 3 + k]) * t; v[j] = (m[j * 3 + i] + m[i * 3 + j]) * t; v[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
    return q;
}
HQ h_qn(HQ a) { const double n = std::sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z); return HQ{a.w / n, a.x / n, a.y / n, a.z / n}; }
HQ h_qinv(HQ a) { const double n2 = a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z; return HQ{a.w / n2, -a.x / n2, -a.y / n2, -a.z / n2}; }
HQ h_qmul(HQ a, HQ b) { return HQ{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x}; }
void h_mm(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j]; C[i * 3 + j] = s; } }
void h_mv(const double *A, const double *v, double *o) { for (int i = 0; i < 3; i++) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2]; }
void h_mtv(const double *A, const double *v, double *o) { for (int i = 0; i < 3; i++) o[i] = A[i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2]; }
void h_log(HQ q, double *om) {   // Sophus SO3::log of a unit quaternion
    const double n2 = q.x * q.x + q.y * q.y + q.z * q.z, n = std::sqrt(n2);
    double f;
    if (n2 < 1e-20) f = 2.0 / q.w - 2.0 / 3.0 * n2 / (q.w * q.w * q.w);
    else if (std::fabs(q.w) < 1e-10) f = (q.w > 0 ? M_PI : -M_PI) / n;
    else f = 2.0 * std::atan(n / q.w) / n;
    om[0] = f * q.x; om[1] = f * q.y; om[2] = f * q.z;
}
HQ h_exp(const double *om) {
    const double t2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2], t = std::sqrt(t2);
    double im, re;
    if (t2 < 1e-20) { const double t4 = t2 * t2; im = 0.5 - t2 / 48.0 + t4 / 3840.0; re = 1.0 - t2 / 8.0 + t4 / 384.0; }
    else { im = std::sin(0.5 * t) / t; re = std::cos(0.5 * t); }
    return HQ{re, im * om[0], im * om[1], im * om[2]};
}
void h_R2ypr(const double *R, double *ypr) {     // Utility::R2ypr, degrees
    const double y = std::atan2(R[3], R[0]), p = std::atan2(-R[6], R[0] * std::cos(y) + R[3] * std::sin(y));
    const double r = std::atan2(R[2] * std::sin(y) - R[5] * std::cos(y), -R[1] * std::sin(y) + R[4] * std::cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;
}
// RelativePoseFactor::update, solver overload (include/factor/relative_pose_factor.h:103-117)
void h_relpose_update(isv_relpose_t *f, const double *ti, const double *Ri, const double *tj, const double *Rj, const double *PSi, const double *PSj) {
    const HQ Qi{PSi[6], PSi[3], PSi[4], PSi[5]}, Qj{PSj[6], PSj[3], PSj[4], PSj[5]};
    double d_tj[3], d_ti[3], Qm[9], A[9], B[9], lgi[3], lgj[3], v1[3], v2[3];
    for (int k = 0; k < 3; k++) { d_tj[k] = PSj[k] - tj[k]; d_ti[k] = PSi[k] - ti[k]; }
    h_q2R(h_qinv(Qj), Qm); h_mm(Qm, Rj, A); h_log(h_qn(h_R2q(A)), lgj);
    h_q2R(h_qinv(Qi), Qm); h_mm(Qm, Ri, B); h_log(h_qn(h_R2q(B)), lgi);
    h_mtv(Ri, d_tj, v1); h_mtv(Ri, d_ti, v2);
    const double *t = f->delta_t;
    const double v3[3] = {t[1] * lgi[2] - t[2] * lgi[1], t[2] * lgi[0] - t[0] * lgi[2], t[0] * lgi[1] - t[1] * lgi[0]};     // skew(delta_t) * log
    for (int k = 0; k < 3; k++) f->delta_t[k] += v1[k] - v2[k] + v3[k];
    double Ji[9], w[3], E[9], T[9];
    h_q2R(h_qmul(h_qinv(Qj), Qi), Ji);
    for (int k = 0; k < 9; k++) Ji[k] = -Ji[k];
    h_mv(Ji, lgi, w);
    h_q2R(h_exp(w), E); h_mm(f->delta_R, E, T); memcpy(f->delta_R, T, sizeof(T));
    h_q2R(h_exp(lgj), E); h_mm(f->delta_R, E, T); memcpy(f->delta_R, T, sizeof(T));
}
void h_inv6(const double *Ain, double *Inv) {    // PartialPivLU inverse of a 6x6 (Eigen MatrixXd::inverse)
    double A[36]; memcpy(A, Ain, sizeof(A));
    int perm[6]; for (int i = 0; i < 6; i++) perm[i] = i;
    for (int k = 0; k < 6; k++) {
        int p = k; double best = std::fabs(A[k * 6 + k]);
        for (int i = k + 1; i < 6; i++) if (std::fabs(A[i * 6 + k]) > best) { best = std::fabs(A[i * 6 + k]); p = i; }
        if (p != k) { for (int j = 0; j < 6; j++) std::swap(A[k * 6 + j], A[p * 6 + j]); std::swap(perm[k], perm[p]); }
        for (int i = k + 1; i < 6; i++) { A[i * 6 + k] /= A[k * 6 + k]; for (int j = k + 1; j < 6; j++) A[i * 6 + j] -= A[i * 6 + k] * A[k * 6 + j]; }
    }
    for (int c = 0; c < 6; c++) {
        double x[6];
        for (int i = 0; i < 6; i++) x[i] = perm[i] == c ? 1.0 : 0.0;
        for (int i = 0; i < 6; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= A[i * 6 + k] * x[k]; x[i] = s; }
        for (int i = 5; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < 6; k++) s -= A[i * 6 + k] * x[k]; x[i] = s / A[i * 6 + i]; }
        for (int i = 0; i < 6; i++) Inv[i * 6 + c] = x[i];
    }
}
}  // namespace

// CombinedFactors::operator+ (include/factor/pose_graph_factors.h:27-51); host arithmetic on 6x6 matrices
extern "C" int isv_combined_factors_add(isv_combined_factors_t *acc, int32_t *acc_length, int64_t *acc_vio_index,
                                        const isv_combined_factors_t *other, int64_t other_vio_index) {
    if (!acc || !acc_length || !acc_vio_index || !other) return ISV_ERR_INVALID_ARG;
    const double *R0 = acc->relative_pose.delta_R, *t0 = acc->relative_pose.delta_t;
    const double *R1 = other->relative_pose.delta_R, *t1 = other->relative_pose.delta_t, *S = other->relative_pose.sqrt_info;
    double W[36], cov1[36], Adj[36] = {0}, T[36], info[36];
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { double s = 0; for (int k = 0; k < 6; k++) s += S[k * 6 + a] * S[k * 6 + b]; W[a * 6 + b] = s; }
    h_inv6(W, cov1);
    // Sophus::SE3d::Adj(): [[R, [t]x R], [0, R]]
    const double Sk[9] = {0, -t0[2], t0[1], t0[2], 0, -t0[0], -t0[1], t0[0], 0};
    double SR[9]; h_mm(Sk, R0, SR);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { Adj[a * 6 + b] = R0[a * 3 + b]; Adj[a * 6 + 3 + b] = SR[a * 3 + b]; Adj[(3 + a) * 6 + 3 + b] = R0[a * 3 + b]; }
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { double s = 0; for (int k = 0; k < 6; k++) s += Adj[a * 6 + k] * cov1[k * 6 + b]; T[a * 6 + b] = s; }
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { double s = 0; for (int k = 0; k < 6; k++) s += T[a * 6 + k] * Adj[b * 6 + k]; acc->covRel[a * 6 + b] += s; }
    acc->has_rollpitch = other->has_rollpitch; acc->rollpitch = other->rollpitch;
    double Rn[9], tn[3], v[3];
    h_mm(R0, R1, Rn); h_mv(R0, t1, v);
    for (int k = 0; k < 3; k++) tn[k] = v[k] + t0[k];
    memcpy(acc->relative_pose.delta_R, Rn, sizeof(Rn)); memcpy(acc->relative_pose.delta_t, tn, sizeof(tn));
    h_inv6(acc->covRel, info);
    double Lm[36]; memcpy(Lm, info, sizeof(Lm));                 // LLT (lower, reads the lower triangle), sqrt_info = L^T
    for (int j = 0; j < 6; j++) {
        double dd = Lm[j * 6 + j];
        for (int k = 0; k < j; k++) dd -= Lm[j * 6 + k] * Lm[j * 6 + k];
        Lm[j * 6 + j] = std::sqrt(dd);
        for (int i = j + 1; i < 6; i++) { double s = Lm[i * 6 + j]; for (int k = 0; k < j; k++) s -= Lm[i * 6 + k] * Lm[j * 6 + k]; Lm[i * 6 + j] = s / Lm[j * 6 + j]; }
    }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) acc->relative_pose.sqrt_info[i * 6 + j] = j >= i ? Lm[j * 6 + i] : 0.0;
    acc->distance = std::sqrt(tn[0] * tn[0] + tn[1] * tn[1] + tn[2] * tn[2]);
    (*acc_length)++;
    if (*acc_vio_index == -1) { memcpy(acc->ti, other->ti, 24); memcpy(acc->Ri, other->Ri, 72); *acc_vio_index = other_vio_index; acc->ts = other->ts; }
    return ISV_OK;
}

extern "C" int isv_pgo_optimize_batch(isv_pgo_t *h, int32_t ng, const int32_t *ns, isv_pg_keyframe_t *const *kfs, const int32_t *firsts,
                                      const int32_t *curs, isv_pgo_result_t *results) {
    if (!h || ng < 1 || !ns || !kfs || !firsts || !curs || !results) return ISV_ERR_INVALID_ARG;
    PCHK(h, hipSetDevice(h->device));
    if (ng > h->cfg.max_graphs) { h->err = "more graphs than max_graphs"; return ISV_ERR_CAPACITY; }
    // host preparation: the graphs are analysed independently (parameter blocks, residual blocks, skyline, column patterns) into
    // per-graph buffers by min(8, cores, graphs) host threads (ISV_HOST_THREADS overrides), then laid end to end
    struct GraphBuild {
        std::vector<double> pose; std::vector<int32_t> free_of, adj_ptr, adj, start, rowptr, colptr, colrows; std::vector<PgEdge> edges; std::vector<uint64_t> raw_key;
        const char *err = nullptr; int rc = ISV_OK;
    };
    std::vector<PgGraph> graphs(ng);
    std::vector<GraphBuild> builds(ng);
    std::vector<std::vector<int>> local(ng);
    std::vector<int> cur_pos(ng, -1), n_loops(ng, 0);
    if (h->cache.size() < (size_t)h->cfg.max_graphs) h->cache.resize((size_t)h->cfg.max_graphs);
    static const bool no_cache = getenv("ISV_PGO_NO_CACHE") != nullptr;       // (A/B and test hook; PROCESS-wide: read once, at the first call)
    std::vector<char> hit(ng, 0);
    auto fill_edge_numbers = [&](PgEdge &E, const isv_pg_keyframe_t &k) {
        if (E.kind == 0) { memcpy(E.meas_R, k.rollpitch.R, 72); memcpy(E.sqrt_info, k.rollpitch.sqrt_info, 32); }
        else if (E.kind == 1) { memcpy(E.meas_t, k.relative_pose.delta_t, 24); memcpy(E.meas_R, k.relative_pose.delta_R, 72); memcpy(E.sqrt_info, k.relative_pose.sqrt_info, 288); }
        else {
            memcpy(E.meas_t, k.loop_info, 24);
            h_q2R(HQ{k.loop_info[3], k.loop_info[4], k.loop_info[5], k.loop_info[6]}, E.meas_R);
            for (int dd = 0; dd < 6; dd++) E.sqrt_info[dd * 6 + dd] = std::sqrt(k.loop_weight);
        }
    };
    auto build_graph = [&](int g) -> int {
        const int n = ns[g]; isv_pg_keyframe_t *kf = kfs[g];
        if (n < 1 || !kf) return ISV_ERR_INVALID_ARG;
        PgGraph &G = graphs[g];
        memset(&G, 0, sizeof(G));
        GraphBuild &B = builds[g];
        // ---- the cached structure of this slot, when nothing it depends on has changed ----
        PgStructCache &SC = h->cache[g];
        uint64_t key = 1469598103934665603ull;
        std::vector<uint64_t> &raw = B.raw_key;
        raw.clear(); raw.reserve(3 + 2 * (size_t)n);
        auto mix = [&](uint64_t v) { raw.push_back(v); key ^= v + 0x9e3779b97f4a7c15ull + (key << 6) + (key >> 2); };
        mix((uint64_t)n); mix((uint64_t)(uint32_t)firsts[g]); mix((uint64_t)(uint32_t)curs[g]);
        for (int k = 0; k < n; k++) {
            mix(((uint64_t)(uint32_t)kf[k].index << 32) | (uint32_t)kf[k].sequence);
            mix(((uint64_t)(kf[k].has_rollpitch != 0) << 33) | ((uint64_t)(kf[k].has_loop != 0) << 32) | (uint32_t)(kf[k].has_loop ? kf[k].loop_index : 0));
        }
        // (the hash only short-cuts the comparison: the tuple itself decides)
        if (!no_cache && SC.valid && SC.key == key && (int)SC.loc.size() == n && SC.raw.size() == raw.size() && memcmp(SC.raw.data(), raw.data(), raw.size() * sizeof(uint64_t)) == 0) {
            G.max_iter = h->cfg.max_iterations; G.huber = h->cfg.huber_delta;
            G.P1 = SC.P1; G.nf = SC.nf; G.nblk = SC.nblk; G.ne = (int32_t)SC.edges.size();
            local[g] = SC.loc; cur_pos[g] = SC.cur_pos; n_loops[g] = SC.n_loops;
            B.pose.reserve((size_t)SC.P1 * 7);
            for (int k = 0; k < n; k++) {
                if (SC.loc[k] < 0) continue;
                for (int c = 0; c < 9; c++) if (!(kf[k].vio_R_w_i[c] - kf[k].vio_R_w_i[c] == 0.0)) { B.err = "non-finite keyframe pose"; return ISV_ERR_NONFINITE; }
                const HQ q = h_qn(h_R2q(kf[k].vio_R_w_i));
                B.pose.insert(B.pose.end(), {kf[k].vio_T_w_i[0], kf[k].vio_T_w_i[1], kf[k].vio_T_w_i[2], q.x, q.y, q.z, q.w});
            }
            B.free_of = SC.free_of; B.adj_ptr = SC.adj_ptr; B.adj = SC.adj; B.start = SC.start; B.rowptr = SC.rowptr; B.colptr = SC.colptr; B.colrows = SC.colrows;
            B.edges.resize(SC.edges.size());
            for (size_t e = 0; e < SC.edges.size(); e++) {
                const PgEdgeTpl &T = SC.edges[e];
                PgEdge &E = B.edges[e];
                memset(&E, 0, sizeof(E));
                E.kind = T.kind; E.a = T.a; E.b = T.b; E.fa = T.fa; E.fb = T.fb; E.dim = T.dim; E.robust = T.robust;
                fill_edge_numbers(E, kf[T.src]);
            }
            hit[g] = 1;
            return ISV_OK;
        }
        SC.valid = false;
        std::vector<int32_t> edge_src;
        std::vector<double> &pose = B.pose; std::vector<int32_t> &free_of = B.free_of, &adj_ptr = B.adj_ptr, &adj = B.adj, &start = B.start, &rowptr = B.rowptr, &colptr = B.colptr, &colrows = B.colrows;
        std::vector<PgEdge> &edges = B.edges;
        G.max_iter = h->cfg.max_iterations; G.huber = h->cfg.huber_delta;
        // parameter blocks: keyframes first_looped_index .. cur_index in list order (pose_graph.cpp:277-306)
        std::vector<int> &loc = local[g];
        loc.assign(n, -1);
        int pi = 0;
        for (int k = 0; k < n; k++) {
            if (kf[k].index < firsts[g] || cur_pos[g] >= 0) continue;
            loc[k] = pi++;
            if (kf[k].index == curs[g]) cur_pos[g] = k;
        }
        if (cur_pos[g] < 0) { B.err = "cur_index is not in the keyframe list (or lies before first_looped_index)"; return ISV_ERR_INVALID_ARG; }
        if (pi > h->cfg.max_keyframes) { B.err = "more keyframes than max_keyframes"; return ISV_ERR_CAPACITY; }
        G.P1 = pi;
        const int param_index = pi - 1;
        std::vector<int> fo(pi, -1);
        int nf = 0;
        for (int k = 0; k < n; k++) {
            const int li = loc[k]; if (li < 0) continue;
            for (int c = 0; c < 9; c++) if (!(kf[k].vio_R_w_i[c] - kf[k].vio_R_w_i[c] == 0.0)) { B.err = "non-finite keyframe pose"; return ISV_ERR_NONFINITE; }
            const HQ q = h_qn(h_R2q(kf[k].vio_R_w_i));                    // tmp_q = tmp_r; tmp_q.normalize()
            pose.insert(pose.end(), {kf[k].vio_T_w_i[0], kf[k].vio_T_w_i[1], kf[k].vio_T_w_i[2], q.x, q.y, q.z, q.w});
            const bool constant = kf[k].index == firsts[g] || kf[k].sequence == 0;
            fo[li] = constant ? -1 : nf++;
        }
        G.nf = nf;
        free_of.insert(free_of.end(), fo.begin(), fo.end());
        // residual blocks of the keyframes BEFORE cur (pose_graph.cpp:309-339)
        std::vector<std::vector<int32_t>> adjl(nf);
        std::vector<int> st(nf);
        for (int f = 0; f < nf; f++) st[f] = f;
        auto add_edge = [&](PgEdge &E) {
            E.fa = fo[E.a]; E.fb = E.kind == 0 ? -1 : fo[E.b];
            const int id = (int)edges.size();
            if (E.fa >= 0) adjl[E.fa].push_back((id << 1) | 0);
            if (E.kind != 0 && E.fb >= 0) adjl[E.fb].push_back((id << 1) | 1);
            if (E.kind != 0 && E.fa >= 0 && E.fb >= 0) { const int lo = std::min(E.fa, E.fb), hi = std::max(E.fa, E.fb); st[hi] = std::min(st[hi], lo); }
            edges.push_back(E);
        };
        for (int k = 0; k < n; k++) {
            const int li = loc[k]; if (li < 0 || k == cur_pos[g]) continue;
            if (kf[k].has_rollpitch) {
                PgEdge E; memset(&E, 0, sizeof(E));
                E.kind = 0; E.a = E.b = li; E.dim = 2;
                memcpy(E.meas_R, kf[k].rollpitch.R, 72); memcpy(E.sqrt_info, kf[k].rollpitch.sqrt_info, 32);
                add_edge(E); edge_src.push_back(k);
            }
            if (li + 1 <= param_index) {
                PgEdge E; memset(&E, 0, sizeof(E));
                E.kind = 1; E.a = li; E.b = li + 1; E.dim = 6;
                memcpy(E.meas_t, kf[k].relative_pose.delta_t, 24); memcpy(E.meas_R, kf[k].relative_pose.delta_R, 72); memcpy(E.sqrt_info, kf[k].relative_pose.sqrt_info, 288);
                add_edge(E); edge_src.push_back(k);
            }
            if (kf[k].has_loop) {
                int conn = -1;
                for (int m = 0; m < n; m++) if (kf[m].index == kf[k].loop_index) conn = loc[m];
                if (conn < 0) { B.err = "loop_index outside the optimised range (the reference asserts loop_index >= first_looped_index)"; return ISV_ERR_INVALID_ARG; }
                PgEdge E; memset(&E, 0, sizeof(E));
                E.kind = 2; E.a = conn; E.b = li; E.dim = 6; E.robust = 1;
                memcpy(E.meas_t, kf[k].loop_info, 24);
                h_q2R(HQ{kf[k].loop_info[3], kf[k].loop_info[4], kf[k].loop_info[5], kf[k].loop_info[6]}, E.meas_R);
                for (int dd = 0; dd < 6; dd++) E.sqrt_info[dd * 6 + dd] = std::sqrt(kf[k].loop_weight);
                add_edge(E); edge_src.push_back(k);
                n_loops[g]++;
            }
        }
        G.ne = (int32_t)edges.size();
        // skyline + adjacency + column patterns
        int nb = 0;
        for (int f = 0; f < nf; f++) { adj_ptr.push_back((int32_t)adj.size()); adj.insert(adj.end(), adjl[f].begin(), adjl[f].end()); start.push_back(st[f]); rowptr.push_back(nb); nb += f - st[f] + 1; }
        adj_ptr.push_back((int32_t)adj.size()); rowptr.push_back(nb);
        G.nblk = nb;
        if (nb - 2 * nf > h->cfg.max_loop_blocks) { B.err = "loop closures span more blocks than max_loop_blocks"; return ISV_ERR_CAPACITY; }
        std::vector<std::vector<int32_t>> cp(nf);
        for (int i = 0; i < nf; i++) for (int j = st[i]; j < i; j++) cp[j].push_back(i);
        for (int j = 0; j < nf; j++) { colptr.push_back((int32_t)colrows.size()); colrows.insert(colrows.end(), cp[j].begin(), cp[j].end()); }
        colptr.push_back((int32_t)colrows.size());
        // remember the structure for the next call on this slot
        SC.key = key; SC.raw = raw; SC.loc = loc; SC.cur_pos = cur_pos[g]; SC.n_loops = n_loops[g]; SC.P1 = G.P1; SC.nf = G.nf; SC.nblk = G.nblk;
        SC.free_of = free_of; SC.adj_ptr = adj_ptr; SC.adj = adj; SC.start = start; SC.rowptr = rowptr; SC.colptr = colptr; SC.colrows = colrows;
        SC.edges.resize(edges.size());
        for (size_t e = 0; e < edges.size(); e++) SC.edges[e] = PgEdgeTpl{edges[e].kind, edges[e].a, edges[e].b, edges[e].fa, edges[e].fb, edges[e].dim, edges[e].robust, edge_src[e]};
        SC.valid = true;
        return ISV_OK;
    };
    auto parallel_over_graphs = [&](const std::function<void(int)> &fn) {
        int T = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *ev = getenv("ISV_HOST_THREADS")) T = atoi(ev);
        T = std::max(1, std::min(T, ng));
        if (T == 1) { for (int g = 0; g < ng; g++) fn(g); return; }
        std::atomic<int> next{0};
        std::vector<std::thread> th;
        for (int k = 0; k < T; k++) th.emplace_back([&] { for (int g; (g = next.fetch_add(1)) < ng; ) fn(g); });
        for (auto &x : th) x.join();
    };
    const auto tp0 = std::chrono::steady_clock::now();
    parallel_over_graphs([&](int g) { builds[g].rc = build_graph(g); });
    const auto tp1 = std::chrono::steady_clock::now();
    for (int g = 0; g < ng; g++) if (builds[g].rc != ISV_OK) { if (builds[g].err) h->err = builds[g].err; return builds[g].rc; }
    for (int g = 0; g < ng; g++) h->cache_hits += hit[g];
    // the batch's arrays live in ONE pinned staging area kept with the handle: no zero-initialised 170 MB vectors per call
    // (a third of a 1024-graph call in round 2), and the copies to and from the device run at the pinned rate
    struct SpanD { double *p = nullptr; size_t n = 0; double *data() const { return p; } size_t size() const { return n; } bool empty() const { return n == 0; } double &operator[](size_t i) const { return p[i]; } };
    struct SpanI { int32_t *p = nullptr; size_t n = 0; int32_t *data() const { return p; } size_t size() const { return n; } bool empty() const { return n == 0; } int32_t &operator[](size_t i) const { return p[i]; } };
    struct SpanE { PgEdge *p = nullptr; size_t n = 0; PgEdge *data() const { return p; } size_t size() const { return n; } bool empty() const { return n == 0; } PgEdge &operator[](size_t i) const { return p[i]; } };
    SpanD pose; SpanI free_of, adj_ptr, adj, start, rowptr, colptr, colrows; SpanE edges;
    size_t nblk_tot = 0, nfree_tot = 0;
    // offsets first (serial, integers only), then the per-graph buffers are copied into place by the host threads: the serial
    // append of round 2 moved ~240 KB per graph (the edge records) on ONE thread, half of a 1024-graph call
    struct Off { size_t pose, free_of, adj_ptr, adj, start, rowptr, colptr, colrows, edges; };
    std::vector<Off> off((size_t)ng + 1);
    memset(&off[0], 0, sizeof(Off));
    for (int g = 0; g < ng; g++) {
        PgGraph &G = graphs[g]; const GraphBuild &B = builds[g]; const Off &o = off[g];
        G.pose0 = (int32_t)(o.pose / 7); G.free0 = (int32_t)nfree_tot; G.edge0 = (int32_t)o.edges; G.blk0 = (int32_t)nblk_tot;
        G.adj0 = (int32_t)o.adj; G.col0 = (int32_t)o.colrows; G.vec0 = (int32_t)(6 * nfree_tot);
        off[g + 1] = Off{o.pose + B.pose.size(), o.free_of + B.free_of.size(), o.adj_ptr + B.adj_ptr.size(), o.adj + B.adj.size(), o.start + B.start.size(),
                         o.rowptr + B.rowptr.size(), o.colptr + B.colptr.size(), o.colrows + B.colrows.size(), o.edges + B.edges.size()};
        nblk_tot += (size_t)G.nblk; nfree_tot += (size_t)G.nf;
    }
    SpanD cov;
    isv_pgo_result_t *res_stage = nullptr;
    {
        const Off &e = off[ng];
        auto al = [](size_t b) { return (b + 63) & ~(size_t)63; };
        const size_t b_pose = al(e.pose * 8), b_cov = al(e.pose / 7 * 36 * 8), b_edges = al(e.edges * sizeof(PgEdge)), b_res = al((size_t)ng * sizeof(isv_pgo_result_t));
        const size_t b_i[7] = {al(e.free_of * 4), al(e.adj_ptr * 4), al(e.adj * 4), al(e.start * 4), al(e.rowptr * 4), al(e.colptr * 4), al(e.colrows * 4)};
        size_t need = b_pose + b_cov + b_edges + b_res;
        for (size_t b : b_i) need += b;
        if (need > h->pin_cap) {
            if (h->pin) (void)hipHostFree(h->pin);
            h->pin = nullptr; h->pin_cap = 0;
            PCHK(h, hipHostMalloc(&h->pin, need + need / 4, hipHostMallocDefault));
            h->pin_cap = need + need / 4;
        }
        char *q = (char *)h->pin;
        pose.p = (double *)q; pose.n = e.pose; q += b_pose;
        cov.p = (double *)q; cov.n = e.pose / 7 * 36; q += b_cov;
        edges.p = (PgEdge *)q; edges.n = e.edges; q += b_edges;
        res_stage = (isv_pgo_result_t *)q; q += b_res;
        SpanI *si[7] = {&free_of, &adj_ptr, &adj, &start, &rowptr, &colptr, &colrows};
        const size_t ni[7] = {e.free_of, e.adj_ptr, e.adj, e.start, e.rowptr, e.colptr, e.colrows};
        for (int k = 0; k < 7; k++) { si[k]->p = (int32_t *)q; si[k]->n = ni[k]; q += b_i[k]; }
    }
    {
        const Off &e = off[ng];
        if (e.pose / 7 > h->cap_pose || e.edges > h->cap_edge || nblk_tot > h->cap_blk || e.adj > h->cap_adj || e.colrows > h->cap_col) {
            h->err = "pose graphs exceed the handle's capacity"; return ISV_ERR_CAPACITY;
        }
    }
    PgDev &d = h->d;
    // index arrays of the envelope in LDS when the largest graph's fit into 48 KB
    size_t max_nf = 0, max_cols = 0;
    for (int g = 0; g < ng; g++) {
        max_nf = std::max(max_nf, (size_t)graphs[g].nf);
        max_cols = std::max(max_cols, (size_t)(graphs[g].nblk - graphs[g].nf));
    }
    size_t idx_bytes = (3 * max_nf + 1 + max_cols) * sizeof(int32_t);
    d.idx_lds_rows = d.idx_lds_cols = 0;
    // (ISV_PGO_IDX_GLOBAL: test hook for the path graphs too large for the LDS copies take)
    if (max_nf > 0 && idx_bytes <= 48 * 1024 && !getenv("ISV_PGO_IDX_GLOBAL")) { d.idx_lds_rows = (int32_t)max_nf; d.idx_lds_cols = (int32_t)max_cols; } else idx_bytes = 0;
    h->last_blocks = (double)nblk_tot;
    // ---- the chunked pipeline: [assemble chunk c into the pinned staging -> H2D -> k_pgo -> D2H] on stream c, then write-back ----
    // One team of host threads walks 2 ng tasks in order (an atomic counter): task g < ng copies graph g's arrays into place, the
    // thread that completes a chunk's last graph enqueues the chunk; task ng + g writes graph g back, after the chunk's stream has
    // drained (the first thread to get there waits for it).  Chunks are contiguous graph ranges, so every array of a chunk is ONE
    // range of the batch's arrays and the offsets the kernel uses are the batch's.
    int C = ng >= 64 ? isv_pgo::PG_CHUNKS : 1;
    if (const char *ev = getenv("ISV_PGO_CHUNKS")) { C = atoi(ev); if (C < 1) C = 1; if (C > isv_pgo::PG_CHUNKS) C = isv_pgo::PG_CHUNKS; if (C > ng) C = ng; }
    auto chunk_lo = [&](int c) { return (int)((int64_t)ng * c / C); };
    auto chunk_of = [&](int g) { int c = (int)(((int64_t)g * C) / ng); while (c + 1 < C && g >= chunk_lo(c + 1)) c++; while (c > 0 && g < chunk_lo(c)) c--; return c; };
    std::vector<std::atomic<int>> left((size_t)C), enq((size_t)C), done((size_t)C);
    for (int c = 0; c < C; c++) { left[c].store(chunk_lo(c + 1) - chunk_lo(c)); enq[c].store(0); done[c].store(0); }
    std::atomic<int> first_chunk{-1}, fail_rc{ISV_OK};
    std::mutex err_mu, done_mu[isv_pgo::PG_CHUNKS];
    auto fail = [&](int rc, const std::string &msg) { std::lock_guard<std::mutex> lk(err_mu); if (fail_rc.load() == ISV_OK) { fail_rc.store(rc); h->err = msg; } };
#define TCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(ISV_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); return; } } while (0)
    bool pgo_waves4 = ng <= h->n_cus;
    if (const char *ev = getenv("ISV_PGO_WAVES")) pgo_waves4 = atoi(ev) > 1;
    auto enqueue_chunk = [&](int c) {
        const int g0 = chunk_lo(c), g1 = chunk_lo(c + 1);
        const Off &a = off[g0], &b = off[g1];
        hipStream_t st = h->cstream[c];
        TCHK(hipSetDevice(h->device));
#define UPR(dst, src, lo, hi) do { if ((hi) > (lo)) TCHK(hipMemcpyAsync((dst) + (lo), (src).data() + (lo), sizeof((src)[0]) * ((hi) - (lo)), hipMemcpyHostToDevice, st)); } while (0)
        TCHK(hipMemcpyAsync(d.graphs + g0, graphs.data() + g0, sizeof(PgGraph) * (size_t)(g1 - g0), hipMemcpyHostToDevice, st));     // (pageable, 80 B per graph)
        UPR(d.pose, pose, a.pose, b.pose); UPR(d.free_of, free_of, a.free_of, b.free_of); UPR(d.edges, edges, a.edges, b.edges);
        UPR(d.adj_ptr, adj_ptr, a.adj_ptr, b.adj_ptr); UPR(d.adj, adj, a.adj, b.adj); UPR(d.start, start, a.start, b.start);
        UPR(d.rowptr, rowptr, a.rowptr, b.rowptr); UPR(d.colptr, colptr, a.colptr, b.colptr); UPR(d.colrows, colrows, a.colrows, b.colrows);
#undef UPR
        PgDev dv = d; dv.g0 = g0;
        TCHK(hipEventRecord(h->kev[c][0], st));
        // (round 5) a call that leaves CUs idle gives every graph four wavefronts -- the same bits (k_pgo's header); ISV_PGO_WAVES=1 / 4 forces a form
        if (pgo_waves4) hipLaunchKernelGGL(k_pgo<4>, dim3(g1 - g0), dim3(256), idx_bytes, st, dv);
        else hipLaunchKernelGGL(k_pgo<1>, dim3(g1 - g0), dim3(64), idx_bytes, st, dv);
        TCHK(hipGetLastError());
        TCHK(hipEventRecord(h->kev[c][1], st));
        // results into the pinned staging (the caller's arrays are pageable: copied from there after the stream has drained)
        if (b.pose > a.pose) {
            TCHK(hipMemcpyAsync(pose.data() + a.pose, d.pose + a.pose, sizeof(double) * (b.pose - a.pose), hipMemcpyDeviceToHost, st));
            TCHK(hipMemcpyAsync(cov.data() + a.pose / 7 * 36, d.cov + a.pose / 7 * 36, sizeof(double) * (b.pose - a.pose) / 7 * 36, hipMemcpyDeviceToHost, st));
        }
        TCHK(hipMemcpyAsync(res_stage + g0, d.res + g0, sizeof(isv_pgo_result_t) * (size_t)(g1 - g0), hipMemcpyDeviceToHost, st));
        int exp = -1; first_chunk.compare_exchange_strong(exp, c);
    };
    auto assemble = [&](int g) {
        GraphBuild &B = builds[g]; const Off &o = off[g];
#define PUT(v) do { if (!B.v.empty()) memcpy(v.data() + o.v, B.v.data(), sizeof(B.v[0]) * B.v.size()); } while (0)
        PUT(pose); PUT(free_of); PUT(adj_ptr); PUT(adj); PUT(start); PUT(rowptr); PUT(colptr); PUT(colrows); PUT(edges);
#undef PUT
        B = GraphBuild();
    };
    // write back (pose_graph.cpp:362-407): updatePose, updateCov, the update() calls, drift, the keyframes after cur
    auto write_back = [&](int g) {
        const int n = ns[g]; isv_pg_keyframe_t *kf = kfs[g];
        const PgGraph &G = graphs[g];
        isv_pgo_result_t &R = results[g];
        R.n_loop_edges = n_loops[g];
        const int param_index = G.P1 - 1;
        isv_pg_keyframe_t *last = nullptr; const double *last_pose = nullptr;
        for (int k = 0; k < n; k++) {
            const int li = local[g][k]; if (li < 0) continue;
            const double *p = pose.data() + (size_t)(G.pose0 + li) * 7;
            memcpy(kf[k].T_w_i, p, 24); h_q2R(HQ{p[6], p[3], p[4], p[5]}, kf[k].R_w_i);
            // (a graph whose covariance factorisation failed -- status ISV_ERR_NONFINITE -- keeps its keyframes' cov /
            // cov_computed as they were: the reference does not check ceres::Covariance::Compute's result and would
            // store whatever GetCovarianceBlock left; zeros marked "computed" are worse than no covariance)
            if (li < param_index && R.status == ISV_OK) {
                // ceres::Covariance::GetCovarianceBlock returns the 7x7 AMBIENT block [Sigma 0; 0 0]; the reference receives it in a
                // 36-double buffer and maps that as a column-major 6x6 (pose_graph.cpp:346-350): reproduced, stored row-major
                double c7[49] = {0};
                const double *S6 = cov.data() + (size_t)(G.pose0 + li) * 36;
                for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) c7[a * 7 + b] = S6[a * 6 + b];
                for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) kf[k].cov[a * 6 + b] = c7[a + 6 * b];
                kf[k].cov_computed = 1;
            }
            if (last) h_relpose_update(&last->relative_pose, last->T_w_i, last->R_w_i, kf[k].T_w_i, kf[k].R_w_i, last_pose, p);
            last = &kf[k]; last_pose = p;
        }
        const isv_pg_keyframe_t &c = kf[cur_pos[g]];
        double yc[3], yv[3], vT[9], t[3];
        h_R2ypr(c.R_w_i, yc); h_R2ypr(c.vio_R_w_i, yv);
        R.yaw_drift = yc[0] - yv[0];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) vT[a * 3 + b] = c.vio_R_w_i[b * 3 + a];
        h_mm(c.R_w_i, vT, R.r_drift);
        h_mv(R.r_drift, c.vio_T_w_i, t);
        for (int k = 0; k < 3; k++) R.t_drift[k] = c.T_w_i[k] - t[k];
        for (int k = cur_pos[g] + 1; k < n; k++) {
            double Pn[3], Rn[9];
            h_mv(R.r_drift, kf[k].vio_T_w_i, Pn); for (int dd = 0; dd < 3; dd++) Pn[dd] += R.t_drift[dd];
            h_mm(R.r_drift, kf[k].vio_R_w_i, Rn);
            memcpy(kf[k].T_w_i, Pn, 24); memcpy(kf[k].R_w_i, Rn, 72);
        }
    };
    std::chrono::steady_clock::time_point tp2 = tp1, tp3 = tp1;      // (trace: the last chunk's enqueue, the last chunk's drain)
    std::chrono::steady_clock::time_point t_enq[isv_pgo::PG_CHUNKS], t_done[isv_pgo::PG_CHUNKS];
    {
        int T = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *ev = getenv("ISV_HOST_THREADS")) T = atoi(ev);
        T = std::max(1, std::min(T, ng));
        std::atomic<int> next{0};
        auto worker = [&] {
            for (int id; (id = next.fetch_add(1)) < 2 * ng; ) {
                if (id < ng) {
                    const int g = id, c = chunk_of(g);
                    if (fail_rc.load() == ISV_OK) assemble(g);
                    if (left[c].fetch_sub(1) == 1) {
                        if (fail_rc.load() == ISV_OK) enqueue_chunk(c);
                        t_enq[c] = std::chrono::steady_clock::now();
                        if (c == C - 1) tp2 = std::chrono::steady_clock::now();
                        enq[c].store(1);
                    }
                } else {
                    const int g = id - ng, c = chunk_of(g);
                    if (!done[c].load()) {
                        std::lock_guard<std::mutex> lk(done_mu[c]);
                        if (!done[c].load()) {
                            while (!enq[c].load()) std::this_thread::yield();
                            if (fail_rc.load() == ISV_OK) {
                                const hipError_t e1 = hipSetDevice(h->device), e2 = e1 == hipSuccess ? hipStreamSynchronize(h->cstream[c]) : e1;
                                if (e2 != hipSuccess) fail(ISV_ERR_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e2));
                                else memcpy(results + chunk_lo(c), res_stage + chunk_lo(c), sizeof(isv_pgo_result_t) * (size_t)(chunk_lo(c + 1) - chunk_lo(c)));
                            }
                            t_done[c] = std::chrono::steady_clock::now();
                            if (c == C - 1) tp3 = std::chrono::steady_clock::now();
                            done[c].store(1);
                        }
                    }
                    if (fail_rc.load() == ISV_OK) write_back(g);
                }
            }
        };
        if (T == 1) worker();
        else {
            std::vector<std::thread> th;
            for (int k = 0; k < T; k++) th.emplace_back(worker);
            for (auto &x : th) x.join();
        }
    }
#undef TCHK
    if (fail_rc.load() != ISV_OK) {
        for (int c = 0; c < C; c++) (void)hipStreamSynchronize(h->cstream[c]);      // nothing of this call stays in flight
        return fail_rc.load();
    }
    {
        // the span of the chunks' kernels: the chunk enqueued first starts first
        float best = 0;
        const int fc = first_chunk.load() < 0 ? 0 : first_chunk.load();
        for (int c = 0; c < C; c++) { float f = 0; if (hipEventElapsedTime(&f, h->kev[fc][0], h->kev[c][1]) == hipSuccess) best = std::max(best, f); else (void)hipGetLastError(); }
        h->last_kernel_ms = best;
    }
    if (getenv("ISV_TRACE_HANDOVER")) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "isv pgo batch: %d graphs: analysis %.2f ms, assembly %.2f ms, H2D + kernel + D2H %.2f ms (%.1f MB up), write-back %.2f ms\n", ng, ms(tp0, tp1), ms(tp1, tp2), ms(tp2, tp3),
                (pose.size() * 8 + edges.size() * sizeof(PgEdge) + (free_of.size() + adj_ptr.size() + adj.size() + start.size() + rowptr.size() + colptr.size() + colrows.size()) * 4) / 1e6, ms(tp3, std::chrono::steady_clock::now()));
        const int fc = first_chunk.load() < 0 ? 0 : first_chunk.load();
        for (int c = 0; c < C; c++) {       // every chunk's kernel on the first chunk's clock, and when the host enqueued / found it drained
            float a = 0, b = 0;
            (void)hipEventElapsedTime(&a, h->kev[fc][0], h->kev[c][0]); (void)hipEventElapsedTime(&b, h->kev[fc][0], h->kev[c][1]); (void)hipGetLastError();
            fprintf(stderr, "  chunk %d: graphs [%d, %d): enqueued at %.2f ms, k_pgo %.2f .. %.2f ms after the first chunk's start, drained at %.2f ms\n", c, chunk_lo(c), chunk_lo(c + 1),
                    ms(tp0, t_enq[c]), a, b, ms(tp0, t_done[c]));
        }
    }
    // per-graph failures surface in the return value too (every graph has been written back by now; results[g].status says which)
    for (int g = 0; g < ng; g++) if (results[g].status != ISV_OK) { h->err = "pose graph " + std::to_string(g) + ": the covariance factorisation failed (poses written, covariances left untouched)"; return results[g].status; }
    return ISV_OK;
}

extern "C" int isv_pgo_optimize(isv_pgo_t *h, int32_t n, isv_pg_keyframe_t *kf, int32_t first_looped_index, int32_t cur_index, isv_pgo_result_t *result) {
    isv_pg_keyframe_t *kfs[1] = {kf};
    return isv_pgo_optimize_batch(h, 1, &n, kfs, &first_looped_index, &cur_index, result);
}

// ./loop_pose_output.txt (pose_graph.cpp:412-423): `fixed` stream formatting (6 decimals)
extern "C" int isv_pgo_write_loop_pose_output(const char *path, int32_t n, const isv_pg_keyframe_t *kf) {
    if (!path || n < 0 || (n > 0 && !kf)) return ISV_ERR_INVALID_ARG;
    FILE *f = fopen(path, "w");
    if (!f) return ISV_ERR_INVALID_ARG;
    for (int k = 0; k < n; k++) {
        const HQ q = h_R2q(kf[k].R_w_i);
        fprintf(f, "%.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", kf[k].time_stamp, kf[k].T_w_i[0], kf[k].T_w_i[1], kf[k].T_w_i[2], q.w, q.x, q.y, q.z);
    }
    fclose(f);
    return ISV_OK;
}
