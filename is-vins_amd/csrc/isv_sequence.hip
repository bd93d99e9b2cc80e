// isv_sequence.hip -- device-resident sequences (include/isvins_backend.h, "device-resident sequences"; SURVEY.md 8f
// rank 1): a slot of the handle keeps ONE sequence's whole window on the device between frames.  Per frame the host hands
// over the newest frame's propagated state, its feature observations and one (two) IMU record(s); the device does
//   k_seq_slide    Estimator::slideWindow  src/estimator.cpp:1565-1698 for the PREVIOUS solve's marginalisation flag: state
//                  and IMU shift, the rotation of the prior factors with the marginalisation outputs (d.marg never left
//                  the device), slideWindowOld -> FeatureManager::removeBackShiftDepth (feature_manager.cpp:275-313, the
//                  depth re-hosting), slideWindowNew -> removeFront (:335-354), removeFailures (:165-174): one stable
//                  compaction of the track list; then installs the newest state and the new IMU record(s)
//   k_seq_append   FeatureManager::addFeatureAndCheckParallax's list update (:52-76): the caller resolved feature id ->
//                  track (its integer bookkeeping), so an observation names its track's ordinal
//   k_seq_build    the solver's view of the window, i.e. what isv_batch_upload's pack_window builds on the host:
//                  goodFeature() landmarks in list order (CSR), the (host, observer) pair groups by a stable counting
//                  sort, their longest-first schedule over the sweep wavefronts, the factor stream of k_lin_gram
//   (k_imu_prep for the new records, k_triangulate, the solve, marginalisation: the existing kernels)
//   k_seq_writeback  setDepth's results back into the track list; the small result record of the frame
// Everything is per window and deterministic (no atomics on reals, fixed orders): a resident sequence gives bitwise
// the results of the re-upload path (tests/test_gpu_resident.py).  The arithmetic that the host path does in C++
// (the re-hosting of a depth) is written in the same operation order WITHOUT FMA contraction.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include "isv_backend_impl.h"

#define ISV_SEQ_RING 32                 // observation ring of a track (>= ISV_MAX_FRAMES, power of two)
#define SEQ_HDR 8                       // ints of the per-window frame header
enum { FH_PREV = 0, FH_MARGIN = 1, FH_NTRK = 2, FH_NOBS = 3, FH_OBSOFF = 4, FH_NIMU = 5, FH_NLM = 6, FH_NF = 7 };
#define SEQ_STATE 24                    // doubles: Ps 3 | Rs 9 | Vs 3 | Bas 3 | Bgs 3 | header0 | pad 2
#define SEQ_OUT 64                      // doubles of the per-window result record
enum { SEQ_ERR_TRACKS = 1, SEQ_ERR_COUNTS = 2, SEQ_ERR_ROLLPITCH = 4, SEQ_ERR_CAP = 8 };

struct SeqDev {
    int32_t Tcap, _pad;
    int32_t *trk_start, *trk_n, *trk_flag, *trk_slot, *trk_off;      // [B][Tcap] in list order
    double *trk_depth;                                                 // [B][Tcap]
    double *pts;                                                       // [B][Tcap][ISV_SEQ_RING][3] by storage slot
    int32_t *n_tracks;                                                 // [B]
    int32_t *lm_track;                                                 // [landmark capacity] CSR landmark -> track ordinal
    int32_t *f_hdr;                                                    // [B][SEQ_HDR]
    isv_seq_obs_t *f_obs;                                              // newest-frame observations of all windows
    double *f_state;                                                   // [B][SEQ_STATE]
    double *f_imu_in, *f_imu_cov;                                      // [B][2][ISV_IMU_IN], [B][2][225]
    int32_t *f_imu_skip;                                               // [B][2]
    int32_t *imu_sel;                                                  // [2 B] factor index of the uploaded records (-1: none)
    double *out;                                                       // [B][SEQ_OUT]
    int32_t *err;                                                      // [B] consistency flags
    // the window as it enters the solve (states and prior factors): a solve that ends non-finite is rolled back to it
    double *Ps0, *Rs0, *Vs0, *Bas0, *Bgs0;
    isv_se3_prior_t *se30; isv_linear9_t *lin90; isv_relpose_t *relpose0; isv_rollpitch_t *rollpitch0;
};


// ------------------------------------------------------------------------------------------------------------------
// block-wide exclusive scan of one int per thread (256 threads), returns the exclusive prefix and the block total
__device__ inline int block_excl_scan(int v, int *sbuf /* [8] */, int t, int &total) {
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off, 64); if ((t & 63) >= off) x += y; }
    if ((t & 63) == 63) sbuf[t >> 6] = x;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < (t >> 6); k++) base += sbuf[k];
    total = sbuf[0] + sbuf[1] + sbuf[2] + sbuf[3];
    __syncthreads();
    return base + x - v;
}

// the host's removeBackShiftDepth arithmetic (isv_estimator.cpp slide_window: mm / mv / mtv / add / sub / mul), same order, no FMA
#pragma clang fp contract(off)
__device__ inline void mm3(const double *A, const double *B, double *C) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}
__device__ inline void mv3(const double *A, const double *x, double *y) {
    y[0] = A[0] * x[0] + A[1] * x[1] + A[2] * x[2]; y[1] = A[3] * x[0] + A[4] * x[1] + A[5] * x[2]; y[2] = A[6] * x[0] + A[7] * x[1] + A[8] * x[2];
}
__device__ inline void mtv3(const double *A, const double *x, double *y) {
    y[0] = A[0] * x[0] + A[3] * x[1] + A[6] * x[2]; y[1] = A[1] * x[0] + A[4] * x[1] + A[7] * x[2]; y[2] = A[2] * x[0] + A[5] * x[1] + A[8] * x[2];
}
__device__ inline double rehost_depth(const double *R0, const double *P0, const double *R1, const double *P1, const double *uv, double depth, double init_depth) {
    double s[3] = {uv[0] * depth, uv[1] * depth, uv[2] * depth}, w0[3], wp[3], dd[3], pj[3];
    mv3(R0, s, w0);
    wp[0] = w0[0] + P0[0]; wp[1] = w0[1] + P0[1]; wp[2] = w0[2] + P0[2];
    dd[0] = wp[0] - P1[0]; dd[1] = wp[1] - P1[1]; dd[2] = wp[2] - P1[2];
    mtv3(R1, dd, pj);
    return pj[2] > 0 ? pj[2] : init_depth;
}

// Estimator::slideWindow for the previous solve, then the newest state and IMU record(s).  One workgroup per window.
__global__ __launch_bounds__(256) void k_seq_slide(DevBatch d, SeqDev s) {
    // N: the REAL frames (ALL_BUF_SIZE); with a free extrinsic the device arrays carry one more (pseudo-)frame per window: Nd, and one
    // more (dummy, skipped) IMU slot: the strides below are Nd / NI, the window logic runs over N / NIr
    const int w = blockIdx.x, t = threadIdx.x, N = d.Nr, Nd = d.N, Nvo = d.Nvo, NI = Nd - 1, NIr = N - 1;
    const int *hdr = s.f_hdr + (size_t)w * SEQ_HDR;
    const int prev = hdr[FH_PREV];
    if (prev < 0) {                                // no frame for this sequence this step (round 4): its pending slide stays pending
        if (t == 0) { s.imu_sel[(size_t)w * 2] = -1; s.imu_sel[(size_t)w * 2 + 1] = -1; }
        return;
    }
    __shared__ int sbuf[8];
    __shared__ double sR0[9], sP0[3], sR1[9], sP1[3];
    __shared__ double sStage[2048];                // prior structs in transit
    double *Ps = d.Ps + (size_t)w * Nd * 3, *Rs = d.Rs + (size_t)w * Nd * 9, *Vs = d.Vs + (size_t)w * Nd * 3, *Bas = d.Bas + (size_t)w * Nd * 3, *Bgs = d.Bgs + (size_t)w * Nd * 3;
    if (prev != 0) {
        // ---- back_R0 / back_P0 and the frame that becomes frame 0 (with the extrinsic folded in, as slideWindowOld does) ----
        if (t == 0) {
            const double *ric = d.ric + (size_t)w * 9, *tic = d.tic + (size_t)w * 3;
            double tt[3];
            mm3(Rs, ric, sR0); mv3(Rs, tic, tt); for (int k = 0; k < 3; k++) sP0[k] = Ps[k] + tt[k];
            mm3(Rs + 9, ric, sR1); mv3(Rs + 9, tic, tt); for (int k = 0; k < 3; k++) sP1[k] = Ps[3 + k] + tt[k];
        }
        __syncthreads();
        // ---- window states: MARGIN_OLD shifts every frame down by one, MARGIN_SECOND_NEW overwrites frame N-2 with N-1 ----
        {
            double *arr[5] = {Ps, Rs, Vs, Bas, Bgs};
            const int wid[5] = {3, 9, 3, 3, 3};
            double v[5][2];
            for (int a = 0; a < 5; a++) {
                const int n = wid[a] * (N - 1);
                for (int u = 0; u < 2; u++) {
                    const int e = t + 256 * u;
                    v[a][u] = 0;
                    if (prev == 1) { if (e < n) v[a][u] = arr[a][e + wid[a]]; }
                    else if (e < wid[a]) v[a][u] = arr[a][(N - 1) * wid[a] + e];
                }
            }
            __syncthreads();
            for (int a = 0; a < 5; a++) {
                const int n = wid[a] * (N - 1);
                for (int u = 0; u < 2; u++) {
                    const int e = t + 256 * u;
                    if (prev == 1) { if (e < n) arr[a][e] = v[a][u]; }
                    else if (e < wid[a]) arr[a][(N - 2) * wid[a] + e] = v[a][u];
                }
            }
        }
        // ---- IMU factors: MARGIN_OLD drops factor 0 (j <- j + 1); MARGIN_SECOND_NEW keeps 0 .. N-4 (N-3 is re-uploaded merged) ----
        if (prev == 1) {
            for (int j = 0; j + 1 < NIr; j++) {
                const size_t fd = (size_t)w * NI + j, fs = fd + 1;
                double a[3];
                int sk = 0;
                // 64 + 225 + 225 doubles per factor: two trips of 256 threads
                for (int u = 0; u < 3; u++) {
                    const int e = t + 256 * u;
                    a[u] = e < ISV_IMU_IN ? d.imu_in[fs * ISV_IMU_IN + e] : (e < ISV_IMU_IN + 225 ? d.imu_cov[fs * 225 + e - ISV_IMU_IN] : (e < ISV_IMU_IN + 450 ? d.imu_sqrt[fs * 225 + e - ISV_IMU_IN - 225] : 0.0));
                }
                if (t == 0) sk = d.imu_skip[fs];
                __syncthreads();
                for (int u = 0; u < 3; u++) {
                    const int e = t + 256 * u;
                    if (e < ISV_IMU_IN) d.imu_in[fd * ISV_IMU_IN + e] = a[u];
                    else if (e < ISV_IMU_IN + 225) d.imu_cov[fd * 225 + e - ISV_IMU_IN] = a[u];
                    else if (e < ISV_IMU_IN + 450) d.imu_sqrt[fd * 225 + e - ISV_IMU_IN - 225] = a[u];
                }
                if (t == 0) d.imu_skip[fd] = sk;
                __syncthreads();
            }
        }
        // ---- prior factors (MARGIN_OLD with marginalisation outputs, :1607-1645): through LDS, then written back ----
        const isv_marg_result_t &m = d.marg[w];
        if (prev == 1 && m.valid) {
            constexpr int RW = sizeof(isv_relpose_t) / 8, PW = sizeof(isv_rollpitch_t) / 8;
            const int nrel = Nvo - 1, nrp = d.n_rp[w];
            uint64_t *stg = (uint64_t *)sStage;
            // relpose[i] <- relpose[i + 1] (shift()), the last one <- backwardRelativePoseEdgeToAdd at (Nvo-2, Nvo-1)
            const uint64_t *rsrc = (const uint64_t *)(d.relpose + (size_t)w * nrel);
            for (int e = t; e < (nrel - 1) * RW; e += 256) stg[e] = rsrc[e + RW];
            for (int e = t; e < RW; e += 256) stg[(nrel - 1) * RW + e] = ((const uint64_t *)&m.backward_relpose)[e];
            // roll/pitch: [old list, backward_rollpitch @ Nvo-1], every index - 1, those below 0 dropped (in order)
            uint64_t *pstg = stg + nrel * RW;
            const uint64_t *psrc = (const uint64_t *)(d.rollpitch + (size_t)w * d.max_rp);
            for (int e = t; e < nrp * PW; e += 256) pstg[e] = psrc[e];
            for (int e = t; e < PW; e += 256) pstg[nrp * PW + e] = ((const uint64_t *)&m.backward_rollpitch)[e];
            __syncthreads();
            uint64_t *rdst = (uint64_t *)(d.relpose + (size_t)w * nrel);
            for (int e = t; e < nrel * RW; e += 256) rdst[e] = stg[e];
            __syncthreads();
            if (t == 0) {
                for (int i = 0; i < nrel; i++) {
                    isv_relpose_t &f = d.relpose[(size_t)w * nrel + i];
                    if (i < nrel - 1) { f.imu_i -= 1; f.imu_j -= 1; } else { f.imu_i = Nvo - 2; f.imu_j = Nvo - 1; }
                }
                isv_rollpitch_t *lst = (isv_rollpitch_t *)pstg;
                lst[nrp].index = Nvo - 1;
                int o = 0;
                for (int i = 0; i <= nrp; i++) {
                    const int idx = lst[i].index - 1;
                    if (idx < 0) continue;
                    if (o >= d.max_rp) { atomicOr(&s.err[w], SEQ_ERR_ROLLPITCH); break; }
                    isv_rollpitch_t f = lst[i]; f.index = idx;
                    d.rollpitch[(size_t)w * d.max_rp + o++] = f;
                }
                d.n_rp[w] = o;
                isv_se3_prior_t pp = m.forward_pose_prior; pp.index = 0; d.se3[w] = pp;
                isv_linear9_t vb = m.backward_vb; vb.index = Nvo - 1; d.lin9[w] = vb;
            }
            __syncthreads();
        }
        // ---- the track list: one stable compaction of removeBackShiftDepth / removeFront and removeFailures ----
        const int T = s.n_tracks[w];
        const size_t tb = (size_t)w * s.Tcap;
        int kept_before = 0;
        const int fc = N - 1;
        for (int c0 = 0; c0 < T; c0 += 256) {
            const int i = c0 + t;
            int st = 0, n = 0, fl = 0, sl = 0, off = 0, keep = 0;
            double dep = 0;
            if (i < T) {
                st = s.trk_start[tb + i]; n = s.trk_n[tb + i]; fl = s.trk_flag[tb + i]; sl = s.trk_slot[tb + i]; off = s.trk_off[tb + i]; dep = s.trk_depth[tb + i];
                keep = 1;
                double *ring = s.pts + ((size_t)tb + sl) * ISV_SEQ_RING * 3;
                if (prev == 1) {
                    if (st != 0) st--;
                    else {
                        const double *uv = ring + (size_t)(off & (ISV_SEQ_RING - 1)) * 3;
                        const double u3[3] = {uv[0], uv[1], uv[2]};
                        off = (off + 1) & (ISV_SEQ_RING - 1); n--;
                        if (n < 2) keep = 0;
                        else dep = rehost_depth(sR0, sP0, sR1, sP1, u3, dep, d.init_depth);
                    }
                } else {
                    if (st == fc) st--;
                    else if (st + n - 1 >= fc - 1) {
                        const int idx = N - 2 - st;          // erase the observation of frame N-2: a later one (frame N-1) moves down
                        if (idx + 1 < n) {
                            double *a = ring + (size_t)((off + idx) & (ISV_SEQ_RING - 1)) * 3, *b = ring + (size_t)((off + idx + 1) & (ISV_SEQ_RING - 1)) * 3;
                            a[0] = b[0]; a[1] = b[1]; a[2] = b[2];
                        }
                        n--;
                        if (n == 0) keep = 0;
                    }
                }
                if (fl == 2) keep = 0;                          // removeFailures
            }
            int tot;
            const int pos = kept_before + block_excl_scan(keep, sbuf, t, tot);
            if (keep) {
                s.trk_start[tb + pos] = st; s.trk_n[tb + pos] = n; s.trk_flag[tb + pos] = fl; s.trk_slot[tb + pos] = sl; s.trk_off[tb + pos] = off; s.trk_depth[tb + pos] = dep;
            }
            kept_before += tot;
            __syncthreads();
        }
        if (t == 0) s.n_tracks[w] = kept_before;
    }
    __syncthreads();
    if (t == 0 && s.n_tracks[w] != hdr[FH_NTRK]) atomicOr(&s.err[w], SEQ_ERR_TRACKS);
    // ---- the newest frame as processIMU propagated it; this solve's flag and Headers[0] ----
    if (hdr[FH_MARGIN] < 0) {                      // flush (isv_backend_seq_flush): the slide only
        if (t == 0) { s.imu_sel[(size_t)w * 2] = -1; s.imu_sel[(size_t)w * 2 + 1] = -1; }
        return;
    }
    const double *fs = s.f_state + (size_t)w * SEQ_STATE;
    if (t < 3) { Ps[(N - 1) * 3 + t] = fs[t]; Vs[(N - 1) * 3 + t] = fs[12 + t]; Bas[(N - 1) * 3 + t] = fs[15 + t]; Bgs[(N - 1) * 3 + t] = fs[18 + t]; }
    if (t < 9) Rs[(N - 1) * 9 + t] = fs[3 + t];
    if (t == 0) { d.margin_old[w] = hdr[FH_MARGIN]; d.header0[w] = fs[21]; }
    // ---- the new IMU record(s): factor N-2 (and N-3, merged, after MARGIN_SECOND_NEW) ----
    const int nimu = hdr[FH_NIMU];
    for (int r = 0; r < nimu; r++) {
        const size_t fd = (size_t)w * NI + (NIr - nimu + r), fsrc = (size_t)w * 2 + r;
        for (int e = t; e < ISV_IMU_IN + 225; e += 256) {
            if (e < ISV_IMU_IN) d.imu_in[fd * ISV_IMU_IN + e] = s.f_imu_in[fsrc * ISV_IMU_IN + e];
            else d.imu_cov[fd * 225 + e - ISV_IMU_IN] = s.f_imu_cov[fsrc * 225 + e - ISV_IMU_IN];
        }
        if (t == 0) { d.imu_skip[fd] = s.f_imu_skip[fsrc]; s.imu_sel[(size_t)w * 2 + r] = (int32_t)fd; }
    }
    if (t == 0) for (int r = nimu; r < 2; r++) s.imu_sel[(size_t)w * 2 + r] = -1;
    if (d.est_ex) {
        // the extrinsic the last solve left in tic / ric (k_finalize, double2vector :577-586) is the pseudo-frame's state of this one
        // (k_vector2double turns it into para_Ex_Pose's twin); its speed / biases stay zero from the seed
        if (t < 3) Ps[N * 3 + t] = d.tic[(size_t)w * 3 + t];
        if (t < 9) Rs[N * 9 + t] = d.ric[(size_t)w * 9 + t];
    }
}
#pragma clang fp contract(fast)

// the newest frame's observations: an existing track gets one more point, a new track starts at frame N-1
__global__ __launch_bounds__(256) void k_seq_append(DevBatch d, SeqDev s) {
    const int w = blockIdx.x, t = threadIdx.x;
    const int *hdr = s.f_hdr + (size_t)w * SEQ_HDR;
    if (hdr[FH_PREV] < 0) return;
    const int T0 = s.n_tracks[w], nobs = hdr[FH_NOBS];
    const isv_seq_obs_t *obs = s.f_obs + hdr[FH_OBSOFF];
    const size_t tb = (size_t)w * s.Tcap;
    __shared__ int s_new;
    if (t == 0) s_new = 0;
    __syncthreads();
    for (int o = t; o < nobs; o += 256) {
        const isv_seq_obs_t ob = obs[o];
        const int ord = ob.track;
        if (ord < 0 || ord >= s.Tcap) { atomicOr(&s.err[w], SEQ_ERR_CAP); continue; }
        int sl, pos;
        if (ord < T0) {
            sl = s.trk_slot[tb + ord];
            const int n = s.trk_n[tb + ord];
            pos = (s.trk_off[tb + ord] + n) & (ISV_SEQ_RING - 1);
            s.trk_n[tb + ord] = n + 1;
        } else {
            sl = ob.slot; pos = 0;
            if (sl < 0 || sl >= s.Tcap) { atomicOr(&s.err[w], SEQ_ERR_CAP); continue; }
            s.trk_start[tb + ord] = d.Nr - 1; s.trk_n[tb + ord] = 1; s.trk_flag[tb + ord] = 0; s.trk_slot[tb + ord] = sl; s.trk_off[tb + ord] = 0; s.trk_depth[tb + ord] = -1.0;
            atomicMax(&s_new, ord + 1 - T0);
        }
        double *p = s.pts + (((size_t)tb + sl) * ISV_SEQ_RING + pos) * 3;
        p[0] = ob.point[0]; p[1] = ob.point[1]; p[2] = ob.point[2];
    }
    __syncthreads();
    if (t == 0) s.n_tracks[w] = T0 + s_new;
}

// The (host, observer) pair groups of a window's factors (stable counting sort in landmark order), the longest-first schedule of the
// groups over the sweep wavefronts and the factor stream of k_lin_gram -- what pack_window builds on the host (isv_backend.hip), from
// the landmark table pass A of the calling kernel left in LDS (sMeta[l] = host | k << 8, sF0[l] = first factor, window-relative).
// Shared by k_seq_build (device-resident sequences) and k_upload_build (isv_batch_upload, round 5).  host_end: landmarks are hosted in
// frames < host_end (Nvo for goodFeature() landmarks, N for a caller's window).
__device__ __forceinline__ void build_pairs_stream(const DevBatch &d, const int w, const int t, const int Lw, const int Fw, const int F0, const int host_end, const int lcap, int *ldsi) {
    const int N = d.Nr, NP = N * (N - 1) / 2;
    unsigned *sMeta = (unsigned *)ldsi;
    int *sF0 = ldsi + lcap, *sSize = sF0 + lcap, *sOff = sSize + NP + 1, *sOrder = sOff + NP + 1, *sBase = sOrder + NP + 1, *sWave = sBase + NP + 1;
    // ---- pass C: factors sorted by (host, observer) pair, stable in landmark order.  A landmark has at most one factor
    //      per pair, so the rank of its factor inside the pair group is the number of EARLIER landmarks in the group:
    //      one thread walks the landmark list per pair (count, then fill) ----
    // (round 4: the landmark list in four QUARTERS per pair -- item (pair, quarter) counts / fills its quarter in order, a quarter's
    //  first rank is the pair's offset plus the earlier quarters' counts: the same stable order with four times the threads; the
    //  55 one-thread walks over ~280 landmarks were 37 of the kernel's 175 us)
    int *sCnt = sWave + NP + 1;                    // [4][NP + 1]
    for (int p = t; p <= NP; p += 256) sSize[p] = 0;
    auto pair_of = [N](int p, int &hh, int &jj) { hh = 0; int rem = p; while (rem >= N - 1 - hh) { rem -= N - 1 - hh; hh++; } jj = hh + 1 + rem; };
    for (int it = t; it < 4 * NP; it += 256) {
        const int qtr = it / NP, p = it - qtr * NP;
        int hh, jj; pair_of(p, hh, jj);
        int c = 0;
        if (hh < host_end) for (int l = Lw * qtr / 4, le = Lw * (qtr + 1) / 4; l < le; l++) { const unsigned m0 = sMeta[l]; const int h = (int)(m0 & 255), k = (int)(m0 >> 8); c += (h == hh && h + k > jj) ? 1 : 0; }
        sCnt[qtr * (NP + 1) + p] = c;
    }
    __syncthreads();
    for (int p = t; p < NP; p += 256) sSize[p] = sCnt[p] + sCnt[(NP + 1) + p] + sCnt[2 * (NP + 1) + p] + sCnt[3 * (NP + 1) + p];
    __syncthreads();
    if (t == 0) { int a = 0; for (int p = 0; p < NP; p++) { sOff[p] = a; a += sSize[p]; } sOff[NP] = a; }
    __syncthreads();
    int32_t *pg_off = d.pg_off + (size_t)w * (NP + 1);
    for (int p = t; p <= NP; p += 256) pg_off[p] = sOff[p];
    for (int it = t; it < 4 * NP; it += 256) {
        const int qtr = it / NP, p = it - qtr * NP;
        int hh, jj; pair_of(p, hh, jj);
        if (hh >= host_end || sCnt[qtr * (NP + 1) + p] == 0) continue;
        int pos = sOff[p];
        for (int q2 = 0; q2 < qtr; q2++) pos += sCnt[q2 * (NP + 1) + p];
        for (int l = Lw * qtr / 4, le = Lw * (qtr + 1) / 4; l < le; l++) {
            const unsigned m0 = sMeta[l]; const int h = (int)(m0 & 255), k = (int)(m0 >> 8);
            if (h == hh && h + k > jj) d.pg_perm[(size_t)F0 + pos++] = sF0[l] + (jj - hh - 1);
        }
    }
    // ---- pass D: longest group first onto the least loaded sweep wavefront (stable order), as the host packer ----
    for (int p = t; p < NP; p += 256) {
        const int sp = sSize[p];
        int rank = 0;
        for (int q = 0; q < NP; q++) { const int sq = sSize[q]; rank += (sq > sp || (sq == sp && q < p)) ? 1 : 0; }
        sOrder[rank] = p;
    }
    __syncthreads();
    int32_t *soff = d.pg_sched_off + (size_t)w * (ISV_SWEEP_WAVES + 1), *sched = d.pg_sched + (size_t)w * NP, *wst = d.pg_wstart + (size_t)w * (ISV_SWEEP_WAVES + 1);
    __shared__ int sT[3 * ISV_SWEEP_WAVES];        // (LDS: dynamically indexed private arrays live in scratch memory -- 36 us of this kernel)
    if (t == 0) {
        int *load = sT, *cntw = sT + ISV_SWEEP_WAVES, *fill = sT + 2 * ISV_SWEEP_WAVES;
        for (int v = 0; v < ISV_SWEEP_WAVES; v++) { load[v] = 0; cntw[v] = 0; fill[v] = 0; }
        for (int q = 0; q < NP; q++) {
            const int p = sOrder[q];
            int best = 0;
            for (int v = 1; v < ISV_SWEEP_WAVES; v++) if (load[v] < load[best]) best = v;
            load[best] += 1 + 4 * ((sSize[p] + 7) / 8);
            sWave[p] = best; cntw[best]++;
        }
        soff[0] = 0;
        for (int v = 0; v < ISV_SWEEP_WAVES; v++) soff[v + 1] = soff[v] + cntw[v];
        for (int hh = 0, p = 0; hh < N - 1; hh++)
            for (int jj = hh + 1; jj < N; jj++, p++) { const int v = sWave[p]; sched[soff[v] + fill[v]++] = hh | (jj << 8) | (p << 16); }
        // stream offsets: the groups back to back in schedule order (wavefront, then pair)
        int q = 0;
        for (int v = 0; v < ISV_SWEEP_WAVES; v++) {
            wst[v] = q;
            for (int e = soff[v]; e < soff[v + 1]; e++) { const int pp = sched[e] >> 16; sBase[pp] = q; q += sSize[pp]; }
        }
        wst[ISV_SWEEP_WAVES] = q;
    }
    __syncthreads();
    // ---- the factor stream of k_lin_gram ----
    // (four entries per thread in flight: the sorted position's record comes through two dependent global loads)
    for (int i0 = t; i0 < Fw; i0 += 4 * 256) {
        int frel4[4], hj4[4]; size_t q4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = i0 + 256 * u < Fw ? i0 + 256 * u : Fw - 1;
            int lo = 0, hi = NP;                    // the pair group of sorted position idx: sOff[lo] <= idx < sOff[lo + 1]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sOff[mid] <= idx) lo = mid; else hi = mid; }
            const int pp = lo;                      // (the LAST group starting at or before idx: empty groups share its offset)
            int hh, jj; pair_of(pp, hh, jj);
            q4[u] = (size_t)F0 + sBase[pp] + (idx - sOff[pp]);
            hj4[u] = (hh << 16) | (jj << 24);
            frel4[u] = d.pg_perm[(size_t)F0 + idx];
        }
        int lm4[4]; double x4[4], y4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const size_t f = (size_t)F0 + frel4[u]; lm4[u] = d.f_rec[f].lm; x4[u] = d.f_pts_j[2 * f]; y4[u] = d.f_pts_j[2 * f + 1]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (i0 + 256 * u < Fw) {
                const size_t q = q4[u];
                d.pg_rec[2 * q] = lm4[u]; d.pg_rec[2 * q + 1] = (int32_t)((unsigned)frel4[u] | (unsigned)hj4[u]);
                d.pg_pts[2 * q] = x4[u]; d.pg_pts[2 * q + 1] = y4[u];
            }
        }
    }
}

// The solver's view of the window from the track list (= pack_window of isv_backend.hip, on the device).
// Dynamic LDS: sMeta [Lcap] uint32 (host | k << 8) | sF0 [Lcap] int32 | per pair: size, off, order, base, wave [5][NP + 1] int32 | quarter counts [4][NP + 1]
__global__ __launch_bounds__(256) void k_seq_build(DevBatch d, SeqDev s, int lcap) {
    extern __shared__ int ldsi[];
    // (Nd: device frames per window = the stride of the state arrays; N: the real frames -- pairs, schedule, factor stream)
    const int w = blockIdx.x, t = threadIdx.x, Nd = d.N, Nvo = d.Nvo;
    unsigned *sMeta = (unsigned *)ldsi;
    int *sF0 = ldsi + lcap;
    __shared__ int sbuf[8];
    const int *hdr = s.f_hdr + (size_t)w * SEQ_HDR;
    if (hdr[FH_PREV] < 0) return;                  // (no frame: the window contributes no landmarks / factors to this step's batch)
    const int T = s.n_tracks[w];
    const size_t tb = (size_t)w * s.Tcap;
    const int L0 = d.lm_off[w], F0 = d.f_off[w];
    // ---- the window as it enters the solve (k_seq_writeback rolls a non-finite solve back to it) ----
    {
        const size_t o3 = (size_t)w * Nd * 3, o9 = (size_t)w * Nd * 9;
        for (int e = t; e < Nd * 3; e += 256) { s.Ps0[o3 + e] = d.Ps[o3 + e]; s.Vs0[o3 + e] = d.Vs[o3 + e]; s.Bas0[o3 + e] = d.Bas[o3 + e]; s.Bgs0[o3 + e] = d.Bgs[o3 + e]; }
        for (int e = t; e < Nd * 9; e += 256) s.Rs0[o9 + e] = d.Rs[o9 + e];
        constexpr int SW = sizeof(isv_se3_prior_t) / 8, LW = sizeof(isv_linear9_t) / 8, RW = sizeof(isv_relpose_t) / 8, PW = sizeof(isv_rollpitch_t) / 8;
        const uint64_t *a = (const uint64_t *)(d.se3 + w); uint64_t *b = (uint64_t *)(s.se30 + w);
        for (int e = t; e < SW; e += 256) b[e] = a[e];
        a = (const uint64_t *)(d.lin9 + w); b = (uint64_t *)(s.lin90 + w);
        for (int e = t; e < LW; e += 256) b[e] = a[e];
        a = (const uint64_t *)(d.relpose + (size_t)w * (Nvo - 1)); b = (uint64_t *)(s.relpose0 + (size_t)w * (Nvo - 1));
        for (int e = t; e < RW * (Nvo - 1); e += 256) b[e] = a[e];
        a = (const uint64_t *)(d.rollpitch + (size_t)w * d.max_rp); b = (uint64_t *)(s.rollpitch0 + (size_t)w * d.max_rp);
        for (int e = t; e < PW * d.max_rp; e += 256) b[e] = a[e];
    }
    // ---- pass A: goodFeature() landmarks in list order -> CSR ----
    int lbase = 0, obase = 0;
    for (int c0 = 0; c0 < T; c0 += 256) {
        const int i = c0 + t;
        int st = 0, n = 0, good = 0;
        if (i < T) { st = s.trk_start[tb + i]; n = s.trk_n[tb + i]; good = (n >= 2 && st < Nvo) ? 1 : 0; }
        int totl, toto;
        const int li = lbase + block_excl_scan(good, sbuf, t, totl);
        const int oi = obase + block_excl_scan(good ? n : 0, sbuf, t, toto);
        if (good) {
            if (li >= lcap) atomicOr(&s.err[w], SEQ_ERR_CAP);
            else {
                const int l = L0 + li, frel = oi - li, f0 = F0 + frel;
                const int sl = s.trk_slot[tb + i], off = s.trk_off[tb + i];
                const double *ring = s.pts + ((size_t)tb + sl) * ISV_SEQ_RING * 3;
                d.lm_host[l] = st; d.lm_k[l] = n; d.lm_f0[l] = f0;
                d.lm_meta[l] = (uint32_t)st | ((uint32_t)n << 8) | ((uint32_t)frel << 16);
                d.depth[l] = s.trk_depth[tb + i];
                s.lm_track[l] = i;
                sMeta[li] = (unsigned)st | ((unsigned)n << 8); sF0[li] = frel;
                const double *p0 = ring + (size_t)(off & (ISV_SEQ_RING - 1)) * 3;
                d.lm_pts_i[(size_t)l * 3] = p0[0]; d.lm_pts_i[(size_t)l * 3 + 1] = p0[1]; d.lm_pts_i[(size_t)l * 3 + 2] = p0[2];
                // (round 4: four observations' points in flight -- clamped ring reads, then the stores; one dependent memory latency
                //  per observation made this loop 75 of the kernel's 175 us at 512 windows)
                for (int o0 = 1; o0 < n; o0 += 4) {
                    double px[4], py[4], pz[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int o = o0 + u < n ? o0 + u : n - 1;
                        const double *p = ring + (size_t)((off + o) & (ISV_SEQ_RING - 1)) * 3;
                        px[u] = p[0]; py[u] = p[1]; pz[u] = p[2];
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int o = o0 + u;
                        if (o < n) {
                            const size_t f = (size_t)f0 + o - 1;
                            FactorRec rc; rc.lm = l; rc.ij = st | ((st + o) << 8);
                            d.f_rec[f] = rc;
                            d.f_pts_j[f * 2] = px[u]; d.f_pts_j[f * 2 + 1] = py[u]; d.f_pts_z[f] = pz[u];
                        }
                    }
                }
            }
        }
        lbase += totl; obase += toto;
    }
    const int Lw = lbase, Fw = obase - lbase;
    if (t == 0 && (Lw != hdr[FH_NLM] || Fw != hdr[FH_NF] || Lw != d.lm_off[w + 1] - L0 || Fw != d.f_off[w + 1] - F0)) atomicOr(&s.err[w], SEQ_ERR_COUNTS);
    __syncthreads();
    if (Lw > lcap) return;
    build_pairs_stream(d, w, t, Lw, Fw, F0, Nvo, lcap, ldsi);
}

// isv_batch_upload's device half (round 5): the caller's windows arrive as RAW CSR -- start frames (d.lm_host), observation offsets
// (optr: the window's lm_obs_ptr, L + 1 entries at lm_off[w] + w), the observations' points (obs_raw, n_obs x 3 at f_off[w] + lm_off[w])
// and depths -- and this kernel derives what pack_window derived on the host: lm_k / lm_f0 / lm_meta / lm_pts_i, the factor records and
// observing points, then the pair groups, the schedule and the factor stream (build_pairs_stream).  Entry for entry the host packer's
// arrays (tests/test_gpu_upload_build.py: the two paths give bitwise the same solves).
__global__ __launch_bounds__(256) void k_upload_build(DevBatch d, const int32_t *optr, const double *obs_raw, int lcap) {
    extern __shared__ int ldsi[];
    const int w = blockIdx.x, t = threadIdx.x;
    unsigned *sMeta = (unsigned *)ldsi;
    int *sF0 = ldsi + lcap;
    const int L0 = d.lm_off[w], Lw = d.lm_off[w + 1] - L0, F0 = d.f_off[w], Fw = d.f_off[w + 1] - F0;
    const int32_t *op = optr + L0 + w;
    const double *pts = obs_raw + (size_t)(F0 + L0) * 3;
    for (int li = t; li < Lw; li += 256) {
        const int l = L0 + li, st = d.lm_host[l], o0 = op[li], n = op[li + 1] - o0, frel = o0 - li, f0 = F0 + frel;
        d.lm_k[l] = n; d.lm_f0[l] = f0;
        d.lm_meta[l] = (uint32_t)st | ((uint32_t)n << 8) | ((uint32_t)frel << 16);
        sMeta[li] = (unsigned)st | ((unsigned)n << 8); sF0[li] = frel;
        const double *p0 = pts + (size_t)o0 * 3;
        d.lm_pts_i[(size_t)l * 3] = p0[0]; d.lm_pts_i[(size_t)l * 3 + 1] = p0[1]; d.lm_pts_i[(size_t)l * 3 + 2] = p0[2];
        for (int o1 = 1; o1 < n; o1 += 4) {                 // (four observations in flight)
            double px[4], py[4], pz[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int o = o1 + u < n ? o1 + u : n - 1; const double *p = p0 + (size_t)o * 3; px[u] = p[0]; py[u] = p[1]; pz[u] = p[2]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int o = o1 + u;
                if (o < n) {
                    const size_t f = (size_t)f0 + o - 1;
                    FactorRec rc; rc.lm = l; rc.ij = st | ((st + o) << 8);
                    d.f_rec[f] = rc;
                    d.f_pts_j[f * 2] = px[u]; d.f_pts_j[f * 2 + 1] = py[u]; d.f_pts_z[f] = pz[u];
                }
            }
        }
    }
    __syncthreads();
    build_pairs_stream(d, w, t, Lw, Fw, F0, d.Nr, lcap, ldsi);
}
size_t upload_build_lds_bytes(int N, int lcap) { return ((size_t)2 * lcap + 9 * ((size_t)N * (N - 1) / 2 + 1)) * sizeof(int32_t); }
int isv_upload_build_enqueue(DevBatch &d, const int32_t *optr, const double *obs_raw, int lcap, hipStream_t st) {
    const size_t lds = upload_build_lds_bytes(d.Nr, lcap);
    if (lds > 48 * 1024) {
        static std::mutex mtx; static size_t cur[64] = {};
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mtx);
        if (lds > cur[dev & 63]) { if (hipFuncSetAttribute((const void *)k_upload_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISV_ERR_DEVICE; cur[dev & 63] = lds; }
    }
    hipLaunchKernelGGL(k_upload_build, dim3(d.B), dim3(256), lds, st, d, optr, obs_raw, lcap);
    return hipGetLastError() == hipSuccess ? ISV_OK : ISV_ERR_DEVICE;
}

// test hook (ISV_DEBUG_SEQ_FAIL_FRAME=k): the solve of window 0 in the k-th resident frame "ends non-finite"
__global__ void k_seq_poison(DevBatch d) { d.st[0].x_cost = __longlong_as_double(0x7ff8000000000000ll); }

// FeatureManager::setDepth's outputs back into the track list, and the frame's small result record
__global__ __launch_bounds__(256) void k_seq_writeback(DevBatch d, SeqDev s) {
    const int w = blockIdx.x, t = threadIdx.x, N = d.N, Nn = d.Nr;      // N: stride (device frames); Nn: the real frames
    if (ISV_SEQ_IDLE(d, w)) return;
    const size_t tb = (size_t)w * s.Tcap;
    __shared__ int s_fail;
    if (t == 0) s_fail = 0;
    __syncthreads();
    const double fc = d.st[w].x_cost;
    if (!(fc - fc == 0.0)) {
        // the solve ended non-finite: the window goes back to what entered the solve, the track list keeps its depths and
        // flags, there are no marginalisation outputs (the host path's policy, isv_estimator.cpp)
        const size_t o3 = (size_t)w * N * 3, o9 = (size_t)w * N * 9;
        for (int e = t; e < N * 3; e += 256) { d.Ps[o3 + e] = s.Ps0[o3 + e]; d.Vs[o3 + e] = s.Vs0[o3 + e]; d.Bas[o3 + e] = s.Bas0[o3 + e]; d.Bgs[o3 + e] = s.Bgs0[o3 + e]; }
        for (int e = t; e < N * 9; e += 256) d.Rs[o9 + e] = s.Rs0[o9 + e];
        constexpr int SW = sizeof(isv_se3_prior_t) / 8, LW = sizeof(isv_linear9_t) / 8, RW = sizeof(isv_relpose_t) / 8, PW = sizeof(isv_rollpitch_t) / 8;
        const uint64_t *a = (const uint64_t *)(s.se30 + w); uint64_t *b = (uint64_t *)(d.se3 + w);
        for (int e = t; e < SW; e += 256) b[e] = a[e];
        a = (const uint64_t *)(s.lin90 + w); b = (uint64_t *)(d.lin9 + w);
        for (int e = t; e < LW; e += 256) b[e] = a[e];
        a = (const uint64_t *)(s.relpose0 + (size_t)w * (d.Nvo - 1)); b = (uint64_t *)(d.relpose + (size_t)w * (d.Nvo - 1));
        for (int e = t; e < RW * (d.Nvo - 1); e += 256) b[e] = a[e];
        a = (const uint64_t *)(s.rollpitch0 + (size_t)w * d.max_rp); b = (uint64_t *)(d.rollpitch + (size_t)w * d.max_rp);
        for (int e = t; e < PW * d.max_rp; e += 256) b[e] = a[e];
        if (d.est_ex) {                             // (the extrinsic too: the pseudo-frame's copy is the one that entered the solve)
            if (t < 3) d.tic[(size_t)w * 3 + t] = s.Ps0[o3 + Nn * 3 + t];
            if (t < 9) d.ric[(size_t)w * 9 + t] = s.Rs0[o9 + Nn * 9 + t];
        }
        if (t == 0) d.marg[w].valid = 0;
        __syncthreads();
    }
    int nf = 0;
    for (int l = d.lm_off[w] + t; (fc - fc == 0.0) && l < d.lm_off[w + 1]; l += 256) {
        const int i = s.lm_track[l], fl = d.solve_flag[l];
        s.trk_depth[tb + i] = d.depth[l]; s.trk_flag[tb + i] = fl;
        nf += fl == 2;
    }
    if (nf) atomicAdd(&s_fail, nf);
    __syncthreads();
    double *o = s.out + (size_t)w * SEQ_OUT;
    const double *Ps = d.Ps + (size_t)w * N * 3, *Rs = d.Rs + (size_t)w * N * 9, *Vs = d.Vs + (size_t)w * N * 3, *Bas = d.Bas + (size_t)w * N * 3, *Bgs = d.Bgs + (size_t)w * N * 3;
    if (t < 3) { o[t] = Ps[(Nn - 1) * 3 + t]; o[12 + t] = Vs[(Nn - 1) * 3 + t]; o[15 + t] = Bas[(Nn - 1) * 3 + t]; o[18 + t] = Bgs[(Nn - 1) * 3 + t]; o[21 + t] = Ps[t]; o[33 + t] = Ps[3 + t]; }
    if (t < 9) { o[3 + t] = Rs[(Nn - 1) * 9 + t]; o[24 + t] = Rs[t]; o[36 + t] = Rs[9 + t]; }
    if (t == 0) { o[45] = d.marg[w].valid; o[46] = s_fail; o[47] = s.err[w]; }
    if (t < 3) o[48 + t] = d.tic[(size_t)w * 3 + t];
    if (t < 9) o[51 + t] = d.ric[(size_t)w * 9 + t];
}

// the seed's points (track-major, oldest observation first, all windows back to back) into the tracks' rings
__global__ __launch_bounds__(256) void k_seq_seed_points(SeqDev s, const double *pts, const int64_t *win_off) {
    const int w = blockIdx.x, t = threadIdx.x;
    const size_t tb = (size_t)w * s.Tcap;
    const int T = s.n_tracks[w];
    __shared__ int sbuf[8];
    int base = 0;
    for (int c0 = 0; c0 < T; c0 += 256) {
        const int i = c0 + t;
        const int n = i < T ? s.trk_n[tb + i] : 0;
        int tot;
        const int off = base + block_excl_scan(n, sbuf, t, tot);
        if (i < T) {
            const double *src = pts + ((size_t)win_off[w] + off) * 3;
            double *ring = s.pts + ((size_t)tb + s.trk_slot[tb + i]) * ISV_SEQ_RING * 3;
            for (int k = 0; k < n * 3; k++) ring[k] = src[k];
        }
        base += tot;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
struct isv_seq_state {
    SeqDev dv{};
    int enabled = 0, seeded = 0, frames_done = 0;
    size_t obs_cap = 0;
    // pinned staging of the frame inputs / outputs
    int32_t *h_hdr = nullptr; isv_seq_obs_t *h_obs = nullptr; double *h_state = nullptr, *h_imu_in = nullptr, *h_imu_cov = nullptr, *h_out = nullptr;
    int32_t *h_imu_skip = nullptr, *h_flags = nullptr;
};

static isv_seq_state *seq_of(isv_backend *h) { return (isv_seq_state *)h->seq; }

extern "C" int isv_backend_seq_enable(isv_backend_t *h, int32_t tracks_per_window) {
    if (!h || tracks_per_window < 1) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    if (h->seq) return ((isv_seq_state *)h->seq)->dv.Tcap >= tracks_per_window ? ISV_OK : ISV_ERR_CAPACITY;
    isv_seq_state *q = new isv_seq_state();
    h->seq = q; h->seq_free = [](void *p) { delete (isv_seq_state *)p; };
    SeqDev &s = q->dv;
    const size_t B = h->capB, T = (size_t)tracks_per_window;
    s.Tcap = tracks_per_window;
    TRY(dalloc(h, &s.trk_start, B * T)); TRY(dalloc(h, &s.trk_n, B * T)); TRY(dalloc(h, &s.trk_flag, B * T)); TRY(dalloc(h, &s.trk_slot, B * T)); TRY(dalloc(h, &s.trk_off, B * T));
    TRY(dalloc(h, &s.trk_depth, B * T)); TRY(dalloc(h, &s.pts, B * T * ISV_SEQ_RING * 3)); TRY(dalloc(h, &s.n_tracks, B)); TRY(dalloc(h, &s.lm_track, h->capL));
    q->obs_cap = B * T;
    TRY(dalloc(h, &s.f_hdr, B * SEQ_HDR)); TRY(dalloc(h, &s.f_obs, q->obs_cap)); TRY(dalloc(h, &s.f_state, B * SEQ_STATE));
    TRY(dalloc(h, &s.f_imu_in, B * 2 * ISV_IMU_IN)); TRY(dalloc(h, &s.f_imu_cov, B * 2 * 225)); TRY(dalloc(h, &s.f_imu_skip, B * 2)); TRY(dalloc(h, &s.imu_sel, B * 2));
    TRY(dalloc(h, &s.out, B * SEQ_OUT)); TRY(dalloc(h, &s.err, B));
    TRY(halloc(h, &q->h_hdr, B * SEQ_HDR)); TRY(halloc(h, &q->h_obs, q->obs_cap)); TRY(halloc(h, &q->h_state, B * SEQ_STATE));
    TRY(halloc(h, &q->h_imu_in, B * 2 * ISV_IMU_IN)); TRY(halloc(h, &q->h_imu_cov, B * 2 * 225)); TRY(halloc(h, &q->h_imu_skip, B * 2));
    TRY(halloc(h, &q->h_out, B * SEQ_OUT)); TRY(halloc(h, &q->h_flags, h->capL));
    HIPCHK(h, hipMemset(s.err, 0, B * sizeof(int32_t)));
    s.Ps0 = h->Ps0; s.Rs0 = h->Rs0; s.Vs0 = h->Vs0; s.Bas0 = h->Bas0; s.Bgs0 = h->Bgs0;
    s.se30 = h->se30; s.lin90 = h->lin90; s.relpose0 = h->relpose0; s.rollpitch0 = h->rollpitch0;
    q->enabled = 1;
    return ISV_OK;
}

extern "C" int isv_backend_seq_seed(isv_backend_t *h, int32_t n, isv_window_t *const *ws, const int32_t *n_tracks,
                                    const isv_seq_track_t *const *tracks, const double *const *points) {
    if (!h || !h->seq || !ws || !n_tracks || !tracks || !points || n < 1) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    isv_seq_state *q = seq_of(h); SeqDev &s = q->dv;
    q->seeded = 0;
    TRY(isv_batch_upload(h, n, ws));                      // states, IMU records (+ sqrt_info), priors, extrinsic
    hipStream_t st = h->stream;
    const size_t T = (size_t)s.Tcap;
    std::vector<int32_t> a_start(n * T, 0), a_n(n * T, 0), a_flag(n * T, 0), a_slot(n * T, 0), a_off(n * T, 0), a_cnt(n);
    std::vector<double> a_depth(n * T, -1.0), a_pts;
    std::vector<int64_t> a_woff((size_t)n);
    for (int b = 0; b < n; b++) {
        if (n_tracks[b] < 0 || (size_t)n_tracks[b] > T) { h->err = "seed: more tracks than tracks_per_window"; return ISV_ERR_CAPACITY; }
        a_cnt[b] = n_tracks[b];
        size_t po = 0;
        for (int i = 0; i < n_tracks[b]; i++) {
            const isv_seq_track_t &tr = tracks[b][i];
            if (tr.slot < 0 || (size_t)tr.slot >= T || tr.n_obs < 1 || tr.n_obs > ISV_SEQ_RING) { h->err = "seed: bad track"; return ISV_ERR_INVALID_ARG; }
            a_start[b * T + i] = tr.start_frame; a_n[b * T + i] = tr.n_obs; a_flag[b * T + i] = tr.solve_flag; a_slot[b * T + i] = tr.slot; a_depth[b * T + i] = tr.depth;
            po += (size_t)tr.n_obs;
        }
        a_woff[b] = (int64_t)(a_pts.size() / 3);
        a_pts.insert(a_pts.end(), points[b], points[b] + po * 3);
    }
    // (one compact upload + a scatter kernel: a copy per track is 2.7 us of stream time each, 0.2 s per 256 sequences)
    double *d_pts = nullptr; int64_t *d_woff = nullptr;
    HIPCHK(h, hipMalloc(&d_pts, sizeof(double) * std::max<size_t>(a_pts.size(), 3)));
    if (hipMalloc(&d_woff, sizeof(int64_t) * (size_t)n) != hipSuccess) { (void)hipFree(d_pts); h->err = "seed: allocation failed"; return ISV_ERR_DEVICE; }
    auto free_tmp = [&]() { (void)hipFree(d_pts); (void)hipFree(d_woff); };
    if ((!a_pts.empty() && hipMemcpyAsync(d_pts, a_pts.data(), sizeof(double) * a_pts.size(), hipMemcpyHostToDevice, st) != hipSuccess) ||
        hipMemcpyAsync(d_woff, a_woff.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) { free_tmp(); h->err = "seed: copy failed"; return ISV_ERR_DEVICE; }
#define UPV(dst, vec) HIPCHK(h, hipMemcpyAsync(dst, (vec).data(), sizeof((vec)[0]) * (vec).size(), hipMemcpyHostToDevice, st))
    UPV(s.trk_start, a_start); UPV(s.trk_n, a_n); UPV(s.trk_flag, a_flag); UPV(s.trk_slot, a_slot); UPV(s.trk_off, a_off); UPV(s.trk_depth, a_depth); UPV(s.n_tracks, a_cnt);
#undef UPV
    hipLaunchKernelGGL(k_seq_seed_points, dim3(n), dim3(256), 0, st, s, d_pts, d_woff);
    if (hipMemsetAsync(s.err, 0, h->capB * sizeof(int32_t), st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { free_tmp(); h->err = "seed: device error"; return ISV_ERR_DEVICE; }
    free_tmp();                                           // (the vectors above are pageable and go out of scope)
    q->seeded = n;
    return ISV_OK;
}

static void pack_imu_record(const isv_imu_t &im, double *r, double *cov, int32_t *skip) {
    memset(r, 0, sizeof(double) * ISV_IMU_IN);
    memcpy(r + IMU_DP, im.delta_p, 24); memcpy(r + IMU_DQ, im.delta_q, 32); memcpy(r + IMU_DV, im.delta_v, 24);
    memcpy(r + IMU_LBA, im.linearized_ba, 24); memcpy(r + IMU_LBG, im.linearized_bg, 24); r[IMU_DT] = im.sum_dt;
    for (int a = 0; a < 3; a++) for (int bb = 0; bb < 3; bb++) {
        r[IMU_DP_DBA + a * 3 + bb] = im.jacobian[(0 + a) * 15 + 9 + bb];
        r[IMU_DP_DBG + a * 3 + bb] = im.jacobian[(0 + a) * 15 + 12 + bb];
        r[IMU_DQ_DBG + a * 3 + bb] = im.jacobian[(3 + a) * 15 + 12 + bb];
        r[IMU_DV_DBA + a * 3 + bb] = im.jacobian[(6 + a) * 15 + 9 + bb];
        r[IMU_DV_DBG + a * 3 + bb] = im.jacobian[(6 + a) * 15 + 12 + bb];
    }
    memcpy(cov, im.covariance, sizeof(double) * 225);
    *skip = im.sum_dt > 10.0;
}

static bool finite_n(const double *p, size_t n) { for (size_t i = 0; i < n; i++) if (!(p[i] - p[i] == 0.0)) return false; return true; }

extern "C" int isv_backend_seq_frame(isv_backend_t *h, int32_t n, const isv_seq_frame_t *fr, isv_seq_result_t *res,
                                     int32_t *const *solve_flags, isv_marg_result_t *marg) {
    if (!h || !h->seq || !fr || !res || n < 1) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    isv_seq_state *q = seq_of(h); SeqDev &s = q->dv;
    if (n != q->seeded) { h->err = "seq_frame: the batch is not the seeded set of sequences"; return ISV_ERR_INVALID_ARG; }
    static const bool trace = getenv("ISV_TRACE_HANDOVER") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const isv_config_t &c = h->cfg;
    DevBatch &d = h->d; hipStream_t st = h->stream;
    const int N = c.n_frames;
    size_t L = 0, F = 0, O = 0, Fmax = 0, Lmax = 0;
    bool any_marg = false;
    int n_idle = 0;
    for (int b = 0; b < n; b++) {
        const isv_seq_frame_t &f = fr[b];
        if (f.prev_slide == -1) {                  // no frame for this sequence this step: an empty slice of the batch, every kernel skips it
            int32_t *hd0 = q->h_hdr + (size_t)b * SEQ_HDR;
            memset(hd0, 0, sizeof(int32_t) * SEQ_HDR);
            hd0[FH_PREV] = -1; hd0[FH_NTRK] = f.n_tracks; hd0[FH_OBSOFF] = (int32_t)O;
            h->h.lm_off[b] = (int32_t)L; h->h.f_off[b] = (int32_t)F;
            n_idle++;
            continue;
        }
        if (f.n_obs < 0 || (f.n_obs > 0 && !f.obs) || !f.imu || f.n_imu < 1 || f.n_imu > 2 || f.prev_slide < 0 || f.prev_slide > 2 || f.n_landmarks < 0 || f.n_factors < 0) return ISV_ERR_INVALID_ARG;
        if (f.n_landmarks > c.max_landmarks || f.n_factors + f.n_landmarks > c.max_obs) { h->err = "window exceeds capacity"; return ISV_ERR_CAPACITY; }
        if (f.n_factors > 65535) { h->err = "more than 65535 factors in one window"; return ISV_ERR_CAPACITY; }
        if (O + (size_t)f.n_obs > q->obs_cap) { h->err = "seq_frame: more observations than the staging holds"; return ISV_ERR_CAPACITY; }
        if (!finite_n(f.Ps, 3 + 9 + 3 + 3 + 3) || !finite_n((const double *)f.imu, (size_t)f.n_imu * (sizeof(isv_imu_t) / 8))) { h->err = "non-finite input"; return ISV_ERR_NONFINITE; }
        int32_t *hd = q->h_hdr + (size_t)b * SEQ_HDR;
        hd[FH_PREV] = f.prev_slide; hd[FH_MARGIN] = f.margin_old != 0; hd[FH_NTRK] = f.n_tracks; hd[FH_NOBS] = f.n_obs; hd[FH_OBSOFF] = (int32_t)O;
        hd[FH_NIMU] = f.n_imu; hd[FH_NLM] = f.n_landmarks; hd[FH_NF] = f.n_factors;
        if (f.n_obs) memcpy(q->h_obs + O, f.obs, sizeof(isv_seq_obs_t) * (size_t)f.n_obs);
        double *sp = q->h_state + (size_t)b * SEQ_STATE;
        memcpy(sp, f.Ps, 24); memcpy(sp + 3, f.Rs, 72); memcpy(sp + 12, f.Vs, 24); memcpy(sp + 15, f.Bas, 24); memcpy(sp + 18, f.Bgs, 24); sp[21] = f.header0; sp[22] = sp[23] = 0;
        for (int r = 0; r < f.n_imu; r++) pack_imu_record(f.imu[r], q->h_imu_in + ((size_t)b * 2 + r) * ISV_IMU_IN, q->h_imu_cov + ((size_t)b * 2 + r) * 225, q->h_imu_skip + (size_t)b * 2 + r);
        h->h.lm_off[b] = (int32_t)L; h->h.f_off[b] = (int32_t)F;
        L += (size_t)f.n_landmarks; F += (size_t)f.n_factors; O += (size_t)f.n_obs;
        Fmax = std::max(Fmax, (size_t)f.n_factors); Lmax = std::max(Lmax, (size_t)f.n_landmarks);
        any_marg |= f.want_marg != 0;
    }
    if (L > h->capL || F > h->capF) { h->err = "batch exceeds capacity"; return ISV_ERR_CAPACITY; }
    h->h.lm_off[n] = (int32_t)L; h->h.f_off[n] = (int32_t)F;
    // the resident path runs the per-window kernels only (no factor tiles are built): ordinary windows
    if (n_idle == n) { h->err = "seq_frame: no sequence has a frame"; return ISV_ERR_INVALID_ARG; }
    d.B = n; d.Ltot = (int32_t)L; d.Ftot = (int32_t)F; d.n_tiles = 0;
    d.seq_hdr = s.f_hdr;
    d.lg_lcap = (int32_t)((Lmax + 31) / 32 * 32);
    const bool lg_fits = lin_gram_lds_bytes(N, true, c.estimate_extrinsic != 0, LG_WAVES, d.lg_lcap) <= ISV_LDS_PER_CU;
    if (!d.lds_T || Fmax > ISV_FUSED_MAX_FACTORS || !lg_fits || h->hc.legacy_visual || F > (size_t)4096 * n) {
        h->err = "device-resident sequences run the per-window kernels only (N <= 20, <= 8192 factors per window): use the upload path";
        return ISV_ERR_UNSUPPORTED;
    }
    d.fused_visual = 1;
    const auto t1 = std::chrono::steady_clock::now();
#define H2D(dst, src, cnt) HIPCHK(h, hipMemcpyAsync(dst, src, sizeof(*(src)) * (size_t)(cnt), hipMemcpyHostToDevice, st))
    H2D(s.f_hdr, q->h_hdr, (size_t)n * SEQ_HDR); if (O) H2D(s.f_obs, q->h_obs, O); H2D(s.f_state, q->h_state, (size_t)n * SEQ_STATE);
    H2D(s.f_imu_in, q->h_imu_in, (size_t)n * 2 * ISV_IMU_IN); H2D(s.f_imu_cov, q->h_imu_cov, (size_t)n * 2 * 225); H2D(s.f_imu_skip, q->h_imu_skip, (size_t)n * 2);
    H2D(d.lm_off, h->h.lm_off, n + 1); H2D(d.f_off, h->h.f_off, n + 1);
#undef H2D
    // the extrinsic the kernels read is the caller's, every frame: k_finalize leaves R(q(ric)) in d.ric, the re-upload path
    // hands the pristine matrix over again
    if (!c.estimate_extrinsic) {       // (a free extrinsic is CARRIED from solve to solve: tic[0] / ric[0] of double2vector, round 4)
        HIPCHK(h, hipMemcpyAsync(d.tic, h->tic0, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipMemcpyAsync(d.ric, h->ric0, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(k_seq_slide, dim3(n), dim3(256), 0, st, d, s);
    hipLaunchKernelGGL(k_seq_append, dim3(n), dim3(256), 0, st, d, s);
    hipLaunchKernelGGL(k_imu_prep, dim3(2 * n), dim3(64), 0, st, d, s.imu_sel);
    const int NP = N * (N - 1) / 2, lcap = c.max_landmarks > 1 ? c.max_landmarks : 1;
    const size_t lds_build = ((size_t)2 * lcap + 9 * (size_t)(NP + 1)) * sizeof(int32_t);
    hipLaunchKernelGGL(k_seq_build, dim3(n), dim3(256), lds_build, st, d, s, lcap);
    if (L) hipLaunchKernelGGL(k_triangulate, dim3((unsigned)((L + 63) / 64)), dim3(64), 0, st, d);
    HIPCHK(h, hipGetLastError());
    memset(h->last_counts, 0, sizeof(h->last_counts));
    HIPCHK(h, hipMemsetAsync(d.act, 0, sizeof(int32_t) * ISV_MAX_TRACE, st));
    HIPCHK(h, hipEventRecord(h->ev[0], st));
    hipLaunchKernelGGL(k_vector2double, dim3(n), dim3(64), 0, st, d);
    TRY(isv_solver_enqueue(h->d, h->hc, st, h->stream2, h->fj, h->last_counts, nullptr, h->err));
    h->prof_valid = 0;
    {
        const char *ff = getenv("ISV_DEBUG_SEQ_FAIL_FRAME");        // (test hook, read per frame: a pointer compare when unset)
        const int fail_frame = ff ? atoi(ff) : -1;
        if (fail_frame >= 0 && q->frames_done == fail_frame) hipLaunchKernelGGL(k_seq_poison, dim3(1), dim3(1), 0, st, d);
        q->frames_done++;
    }
    hipLaunchKernelGGL(k_seq_writeback, dim3(n), dim3(256), 0, st, d, s);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev[4], st));
    HIPCHK(h, hipMemcpyAsync(q->h_out, s.out, sizeof(double) * (size_t)n * SEQ_OUT, hipMemcpyDeviceToHost, st));
    if (solve_flags && L) HIPCHK(h, hipMemcpyAsync(q->h_flags, d.solve_flag, sizeof(int32_t) * L, hipMemcpyDeviceToHost, st));
    std::vector<isv_summary_t> sums((size_t)n);
    TRY(isv_solver_download(h->d, st, n, h->stage, sums.data(), any_marg ? marg : nullptr, h->err));      // (synchronises the stream)
    const auto t2 = std::chrono::steady_clock::now();
    h->resident = 0;                                      // (the batch API's restore copies do not describe this state)
    int rc = ISV_OK;
    for (int b = 0; b < n; b++) {
        isv_seq_result_t &r = res[b];
        if (fr[b].prev_slide == -1) { memset(&r, 0, sizeof(r)); continue; }      // (no frame: nothing was solved or written back)
        const double *o = q->h_out + (size_t)b * SEQ_OUT;
        r.summary = sums[b];
        memcpy(r.Ps_new, o, 24); memcpy(r.Rs_new, o + 3, 72); memcpy(r.Vs_new, o + 12, 24); memcpy(r.Bas_new, o + 15, 24); memcpy(r.Bgs_new, o + 18, 24);
        memcpy(r.Ps_old, o + 21, 24); memcpy(r.Rs_old, o + 24, 72); memcpy(r.Ps_second, o + 33, 24); memcpy(r.Rs_second, o + 36, 72);
        r.marg_valid = (int32_t)o[45]; r.n_failed_landmarks = (int32_t)o[46];
        memcpy(r.tic, o + 48, 24); memcpy(r.ric, o + 51, 72);
        if ((int)o[47] != 0) { h->err = "seq_frame: the device's track list disagrees with the caller's bookkeeping (window " + std::to_string(b) + ", flags " + std::to_string((int)o[47]) + ")"; rc = ISV_ERR_INVALID_ARG; }
        if (solve_flags && solve_flags[b]) memcpy(solve_flags[b], q->h_flags + h->h.lm_off[b], sizeof(int32_t) * (size_t)fr[b].n_landmarks);
    }
    if (rc != ISV_OK) q->seeded = 0;
    if (trace) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "isv resident frame: n=%d L=%zu F=%zu new obs %zu: pack %.2f ms, device %.2f ms, unpack %.2f ms\n", n, L, F, O, ms(t0, t1), ms(t1, t2), ms(t2, std::chrono::steady_clock::now()));
    }
    return rc;
}

// apply the slide the caller has already made on its side (a sequence leaving the resident mode between two frames)
extern "C" int isv_backend_seq_flush(isv_backend_t *h, int32_t n, const int32_t *prev_slide, const int32_t *n_tracks) {
    if (!h || !h->seq || !prev_slide || !n_tracks || n < 1) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    isv_seq_state *q = seq_of(h); SeqDev &s = q->dv;
    if (n != q->seeded) return ISV_ERR_INVALID_ARG;
    hipStream_t st = h->stream;
    for (int b = 0; b < n; b++) {
        int32_t *hd = q->h_hdr + (size_t)b * SEQ_HDR;
        memset(hd, 0, sizeof(int32_t) * SEQ_HDR);
        hd[FH_PREV] = prev_slide[b]; hd[FH_MARGIN] = -1; hd[FH_NTRK] = n_tracks[b];
    }
    HIPCHK(h, hipMemcpyAsync(s.f_hdr, q->h_hdr, sizeof(int32_t) * (size_t)n * SEQ_HDR, hipMemcpyHostToDevice, st));
    if (!h->cfg.estimate_extrinsic) {
        HIPCHK(h, hipMemcpyAsync(h->d.tic, h->tic0, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipMemcpyAsync(h->d.ric, h->ric0, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToDevice, st));
    }
    h->d.B = n;
    hipLaunchKernelGGL(k_seq_slide, dim3(n), dim3(256), 0, st, h->d, s);
    HIPCHK(h, hipGetLastError());
    std::vector<int32_t> err((size_t)n);
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipMemcpy(err.data(), s.err, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    for (int b = 0; b < n; b++) if (err[b]) { h->err = "seq_flush: the device's track list disagrees with the caller's bookkeeping"; q->seeded = 0; return ISV_ERR_INVALID_ARG; }
    return ISV_OK;
}

extern "C" int isv_backend_seq_marg(isv_backend_t *h, int32_t slot, isv_marg_result_t *out) {
    if (!h || !h->seq || !out || slot < 0 || (size_t)slot >= h->capB) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, h->d.marg + slot, sizeof(isv_marg_result_t), hipMemcpyDeviceToHost));
    return ISV_OK;
}

extern "C" int isv_backend_seq_download(isv_backend_t *h, int32_t slot, isv_window_t *w, int32_t n_tracks, double *track_depth, int32_t *track_flag) {
    if (!h || !h->seq || !w || slot < 0 || (size_t)slot >= h->capB) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    isv_seq_state *q = seq_of(h); SeqDev &s = q->dv; DevBatch &d = h->d; const isv_config_t &c = h->cfg;
    const size_t N = d.N, b = (size_t)slot;
    HIPCHK(h, hipStreamSynchronize(h->stream));
#define DH(dst, src, cnt) HIPCHK(h, hipMemcpy(dst, src, sizeof(*(src)) * (size_t)(cnt), hipMemcpyDeviceToHost))
    const size_t Nr = d.Nr;                    // (the caller's arrays hold the real frames; N is the device stride)
    DH(w->Ps, d.Ps + b * N * 3, Nr * 3); DH(w->Rs, d.Rs + b * N * 9, Nr * 9); DH(w->Vs, d.Vs + b * N * 3, Nr * 3); DH(w->Bas, d.Bas + b * N * 3, Nr * 3); DH(w->Bgs, d.Bgs + b * N * 3, Nr * 3);
    if (w->tic) DH(w->tic, d.tic + b * 3, 3);
    if (w->ric) DH(w->ric, d.ric + b * 9, 9);
    DH(w->pose_prior, d.se3 + b, 1); DH(w->vb_prior, d.lin9 + b, 1); DH(w->relpose, d.relpose + b * (c.n_vo - 1), c.n_vo - 1);
    int32_t nrp = 0;
    DH(&nrp, d.n_rp + b, 1);
    w->n_rollpitch = nrp;
    if (nrp > 0 && w->rollpitch) DH(w->rollpitch, d.rollpitch + b * c.max_rollpitch, nrp);
    if (n_tracks > 0) {
        if (n_tracks > s.Tcap) return ISV_ERR_CAPACITY;
        if (track_depth) DH(track_depth, s.trk_depth + b * s.Tcap, n_tracks);
        if (track_flag) DH(track_flag, s.trk_flag + b * s.Tcap, n_tracks);
    }
#undef DH
    return ISV_OK;
}
