// isv_backend.hip -- host side of the C ABI (include/isvins_backend.h): owns the device-resident
// batch, packs caller windows into it, and enqueues the kernel sequence on the handle's HIP stream.
// No torch types, no allocation across the ABI.  There is NO CPU fallback: every entry point
// fails with ISV_ERR_DEVICE when HIP is unavailable.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <sched.h>
#include <vector>
#include <algorithm>
#include "isv_device_types.h"
#include "isv_kernels.h"

#include "isv_backend_impl.h"

extern "C" int isv_abi_version(void) { return ISV_ABI_VERSION; }

extern "C" const char *isv_backend_last_error(const isv_backend_t *h) { return h ? h->err.c_str() : "null handle"; }

extern "C" void isv_backend_destroy(isv_backend_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->seq && h->seq_free) h->seq_free(h->seq);
    if (h->init_scratch) (void)hipFree(h->init_scratch);
    if (h->init_kld) (void)hipFree(h->init_kld);
    for (void *p : h->allocs) (void)hipFree(p);
    for (void *p : h->hallocs) (void)hipHostFree(p);
    for (auto &e : h->ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : h->prof_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : h->fj) if (e) (void)hipEventDestroy(e);
    for (auto &e : h->pk) if (e) (void)hipEventDestroy(e);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

static int create_impl(isv_backend *h) {
    const isv_config_t &c = h->cfg;
    int ndev = 0;
    HIPCHK(h, hipGetDeviceCount(&ndev));
    if (ndev <= 0) { h->err = "no HIP device"; return ISV_ERR_DEVICE; }
    HIPCHK(h, hipGetDevice(&h->device));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    for (auto &e : h->fj) HIPCHK(h, hipEventCreateWithFlags(&e, getenv("ISV_EVENT_SYSTEM") ? hipEventDisableTiming : (hipEventDisableTiming | hipEventReleaseToDevice)));
    for (auto &e : h->pk) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : h->ev) HIPCHK(h, hipEventCreate(&e));
    h->prof_ev.assign((size_t)c.num_iterations * ISV_PROF_FAMILIES * 2, nullptr);
    for (auto &e : h->prof_ev) HIPCHK(h, hipEventCreate(&e));
    // (N: DEVICE frames = ALL_BUF_SIZE, plus the extrinsic's pseudo-frame when it is estimated -- isv_device_types.h)
    const size_t B = c.max_batch, N = (size_t)c.n_frames + (c.estimate_extrinsic ? 1 : 0), L = B * (size_t)c.max_landmarks, F = B * (size_t)c.max_obs;
    const size_t T = F / (ISV_TILE / 2) + B + 1;      // tiles hold whole landmarks: >= 33 factors each when N <= 32
    h->capB = B; h->capL = L; h->capF = F; h->capTiles = T;
    DevBatch &d = h->d;
    d.N = (int32_t)N; d.Nr = c.n_frames; d.est_ex = c.estimate_extrinsic ? 1 : 0; d.Nvo = c.n_vo; d.np = 15 * (int32_t)N; d.max_rp = c.max_rollpitch; d.max_iter = c.num_iterations;
    d.n_prior_slots = 2 + (c.n_vo - 1) + c.max_rollpitch;
    d.max_lm = c.max_landmarks > 1 ? c.max_landmarks : 1;
    d.force_retry = getenv("ISV_DEBUG_FORCE_RETRY") ? atoi(getenv("ISV_DEBUG_FORCE_RETRY")) : 0;
    d.force_invalid = getenv("ISV_DEBUG_FORCE_INVALID") ? atoi(getenv("ISV_DEBUG_FORCE_INVALID")) : 0;
    d.min_radius = getenv("ISV_DEBUG_MIN_RADIUS") ? atof(getenv("ISV_DEBUG_MIN_RADIUS")) : 1e-32;
    if (!(d.min_radius > 0)) d.min_radius = 1e-32;
    d.prior_strip_sz = PR_REL0 + PR_REL_SZ * (c.n_vo - 1) + PR_RP_SZ * c.max_rollpitch;
    memcpy(d.proj_sqrt_info, c.proj_sqrt_info, sizeof(d.proj_sqrt_info));
    memcpy(d.G, c.gravity, sizeof(d.G));
    d.alpha_cut = c.alpha;
    d.init_depth = c.init_depth;
    const size_t NI = B * (N - 1);
    TRY(dalloc(h, &d.Ps, B * N * 3)); TRY(dalloc(h, &d.Rs, B * N * 9)); TRY(dalloc(h, &d.Vs, B * N * 3));
    TRY(dalloc(h, &d.Bas, B * N * 3)); TRY(dalloc(h, &d.Bgs, B * N * 3)); TRY(dalloc(h, &d.tic, B * 3)); TRY(dalloc(h, &d.ric, B * 9));
    TRY(dalloc(h, &d.depth, L)); TRY(dalloc(h, &d.solve_flag, L));
    TRY(dalloc(h, &d.pose, B * N * 7)); TRY(dalloc(h, &d.sb, B * N * 9)); TRY(dalloc(h, &d.ex, B * 7)); TRY(dalloc(h, &d.lam, L));
    TRY(dalloc(h, &d.cpose, B * N * 7)); TRY(dalloc(h, &d.csb, B * N * 9)); TRY(dalloc(h, &d.clam, L));
    TRY(dalloc(h, &d.lm_off, B + 1)); TRY(dalloc(h, &d.f_off, B + 1));
    TRY(dalloc(h, &d.lm_host, L)); TRY(dalloc(h, &d.lm_k, L)); TRY(dalloc(h, &d.lm_f0, L)); TRY(dalloc(h, &d.lm_pts_i, L * 3));
    TRY(dalloc(h, &d.f_rec, F)); TRY(dalloc(h, &d.f_pts_j, F * 2)); TRY(dalloc(h, &d.f_pts_z, F));
    TRY(dalloc(h, &d.tile_win, T)); TRY(dalloc(h, &d.tile_f0, T)); TRY(dalloc(h, &d.tile_n, T));
    TRY(dalloc(h, &d.pg_perm, F)); TRY(dalloc(h, &d.pg_off, B * ((size_t)c.n_frames * (c.n_frames - 1) / 2 + 1)));
    TRY(dalloc(h, &d.pg_sched, B * ((size_t)c.n_frames * (c.n_frames - 1) / 2))); TRY(dalloc(h, &d.pg_sched_off, B * (ISV_SWEEP_WAVES + 1)));
    TRY(dalloc(h, &d.pg_rec, F * 2)); TRY(dalloc(h, &d.pg_pts, F * 2)); TRY(dalloc(h, &d.flm, F * 8)); TRY(dalloc(h, &d.pg_wstart, B * (ISV_SWEEP_WAVES + 1)));
    // d.fused_visual (k_lin_gram, or the unfused k_proj_linearize<0> + k_sweep_mfma pair) is decided per UPLOAD from the
    // windows that were handed over, not from the handle's capacity: see isv_batch_upload
    d.fused_visual = 1;
    TRY(dalloc(h, &d.imu_in, NI * ISV_IMU_IN)); TRY(dalloc(h, &d.imu_cov, NI * 225)); TRY(dalloc(h, &d.imu_sqrt, NI * 225));
    TRY(dalloc(h, &d.imu_skip, NI));
    TRY(dalloc(h, &d.se3, B)); TRY(dalloc(h, &d.lin9, B)); TRY(dalloc(h, &d.relpose, B * (c.n_vo - 1))); TRY(dalloc(h, &d.rollpitch, B * (size_t)c.max_rollpitch));
    TRY(dalloc(h, &d.n_rp, B));
    TRY(dalloc(h, &d.strip, F * ISV_PROJ_STRIP)); TRY(dalloc(h, &d.fcost, F));
    TRY(dalloc(h, &d.imu_strip, NI * ISV_IMU_STRIP)); TRY(dalloc(h, &d.imu_cost, NI));
    TRY(dalloc(h, &d.imu_raw, (NI + 7) / 8 * 8 * 144)); HIPCHK(h, hipMemset(d.imu_raw, 0, (NI + 7) / 8 * 8 * 144 * sizeof(double)));      // (ISV_IMU_RAWC = 144, isv_linearize.hip)
    TRY(dalloc(h, &d.prior_strip, B * (size_t)d.prior_strip_sz)); TRY(dalloc(h, &d.prior_cost, B * (size_t)d.n_prior_slots));
    TRY(dalloc(h, &d.cost, B)); TRY(dalloc(h, &d.st, B));
    d.prior_H_sz = PH_REL0 + PH_REL_SZ * (c.n_vo - 1) + PH_RP_SZ * c.max_rollpitch;
    TRY(dalloc(h, &d.imu_H, NI * ISV_IMU_H)); TRY(dalloc(h, &d.prior_H, B * (size_t)d.prior_H_sz));
    HIPCHK(h, hipMemset(d.imu_H, 0, NI * ISV_IMU_H * sizeof(double)));      // (the solve kernels read EVERY record and mask the skipped factors' by multiplication: a record that is never written must be finite)
    TRY(dalloc(h, &d.lm_meta, L));
    TRY(dalloc(h, &h->Ps0, B * N * 3)); TRY(dalloc(h, &h->Rs0, B * N * 9)); TRY(dalloc(h, &h->Vs0, B * N * 3));
    TRY(dalloc(h, &h->Bas0, B * N * 3)); TRY(dalloc(h, &h->Bgs0, B * N * 3)); TRY(dalloc(h, &h->depth0, L));
    TRY(dalloc(h, &h->tic0, B * 3)); TRY(dalloc(h, &h->ric0, B * 9));
    TRY(dalloc(h, &h->se30, B)); TRY(dalloc(h, &h->lin90, B)); TRY(dalloc(h, &h->relpose0, B * (c.n_vo - 1))); TRY(dalloc(h, &h->rollpitch0, B * (size_t)c.max_rollpitch));
    auto &s = h->h;
    TRY(halloc(h, &s.Ps, B * N * 3)); TRY(halloc(h, &s.Rs, B * N * 9)); TRY(halloc(h, &s.Vs, B * N * 3));
    TRY(halloc(h, &s.Bas, B * N * 3)); TRY(halloc(h, &s.Bgs, B * N * 3)); TRY(halloc(h, &s.tic, B * 3)); TRY(halloc(h, &s.ric, B * 9));
    TRY(halloc(h, &s.depth, L)); TRY(halloc(h, &s.solve_flag, L)); TRY(halloc(h, &s.lm_pts_i, L * 3)); TRY(halloc(h, &s.f_pts_j, F * 2)); TRY(halloc(h, &s.f_pts_z, F));
    TRY(halloc(h, &s.imu_in, NI * ISV_IMU_IN)); TRY(halloc(h, &s.imu_cov, NI * 225));
    TRY(halloc(h, &s.lm_off, B + 1)); TRY(halloc(h, &s.f_off, B + 1)); TRY(halloc(h, &s.lm_host, L)); TRY(halloc(h, &s.lm_k, L)); TRY(halloc(h, &s.lm_f0, L));
    TRY(halloc(h, &s.tile_win, T)); TRY(halloc(h, &s.tile_f0, T)); TRY(halloc(h, &s.tile_n, T));
    TRY(halloc(h, &s.pg_perm, F)); TRY(halloc(h, &s.pg_off, B * ((size_t)c.n_frames * (c.n_frames - 1) / 2 + 1)));
    TRY(halloc(h, &s.pg_sched, B * ((size_t)c.n_frames * (c.n_frames - 1) / 2))); TRY(halloc(h, &s.pg_sched_off, B * (ISV_SWEEP_WAVES + 1))); TRY(halloc(h, &s.imu_skip, NI)); TRY(halloc(h, &s.n_rp, B));
    TRY(halloc(h, &s.f_rec, F)); TRY(halloc(h, &s.pg_rec, F * 2)); TRY(halloc(h, &s.pg_pts, F * 2)); TRY(halloc(h, &s.pg_wstart, B * (ISV_SWEEP_WAVES + 1)));
    TRY(halloc(h, &s.margin_old, B)); TRY(halloc(h, &s.header0, B));
    TRY(halloc(h, &s.lm_meta, L));
    TRY(halloc(h, &s.lm_optr, L + B)); TRY(halloc(h, &s.obs_raw, F * 3)); TRY(dalloc(h, &h->d_optr, L + B)); TRY(dalloc(h, &h->d_obs_raw, F * 3));
    h->dev_build = getenv("ISV_HOST_PACK") == nullptr && upload_build_lds_bytes(c.n_frames, c.max_landmarks > 1 ? c.max_landmarks : 1) <= 64 * 1024;
    TRY(halloc(h, &s.se3, B)); TRY(halloc(h, &s.lin9, B)); TRY(halloc(h, &s.relpose, B * (c.n_vo - 1))); TRY(halloc(h, &s.rollpitch, B * (size_t)c.max_rollpitch));
    TRY(halloc(h, &s.st, B));
    TRY(halloc(h, &s.pose, B * N * 7)); TRY(halloc(h, &s.sb, B * N * 9)); TRY(halloc(h, &s.ex, B * 7)); TRY(halloc(h, &s.lam, L));
    TRY(halloc(h, &h->stage.st, B)); TRY(halloc(h, &h->stage.tc, B * ISV_MAX_TRACE)); TRY(halloc(h, &h->stage.tr, B * ISV_MAX_TRACE));
    TRY(halloc(h, &h->stage.ts, B * ISV_MAX_TRACE)); TRY(halloc(h, &h->stage.ta, B * ISV_MAX_TRACE)); TRY(halloc(h, &h->stage.marg, B));
    TRY(isv_solver_alloc(h->d, h->hc, B, L, F, h->allocs, h->err));
    {
        // (round 5) everything a RAW upload sends lives in ONE pinned block and ONE device block with the same layout, so that an upload
        // is ONE host-to-device copy: on this GPU a copy command costs 20-50 us of stream time whatever its size (rocprofv3: 30 commands
        // per upload, 0.7 ms of transfers inside 1.5 ms of stream time).  The arrays were allocated one by one above; they are re-pointed
        // into the blocks and the originals are released.
        struct Item { void **hp, **dp; size_t bytes, off; };
        std::vector<Item> it;
#define ARENA(hptr, dptr, cnt) it.push_back(Item{(void **)&(hptr), (void **)&(dptr), sizeof(*(hptr)) * (size_t)(cnt), 0})
        ARENA(s.lm_off, d.lm_off, B + 1); ARENA(s.f_off, d.f_off, B + 1);
        ARENA(s.Ps, d.Ps, B * N * 3); ARENA(s.Rs, d.Rs, B * N * 9); ARENA(s.Vs, d.Vs, B * N * 3); ARENA(s.Bas, d.Bas, B * N * 3); ARENA(s.Bgs, d.Bgs, B * N * 3);
        ARENA(s.tic, d.tic, B * 3); ARENA(s.ric, d.ric, B * 9);
        ARENA(s.se3, d.se3, B); ARENA(s.lin9, d.lin9, B); ARENA(s.relpose, d.relpose, B * (c.n_vo - 1)); ARENA(s.rollpitch, d.rollpitch, B * (size_t)c.max_rollpitch);
        ARENA(s.depth, d.depth, L);            // (Ps .. depth: what isv_batch_download brings back from this block, in one copy)
        ARENA(s.n_rp, d.n_rp, B); ARENA(s.margin_old, d.margin_old, B); ARENA(s.header0, d.header0, B);
        ARENA(s.imu_skip, d.imu_skip, NI); ARENA(s.imu_in, d.imu_in, NI * ISV_IMU_IN); ARENA(s.imu_cov, d.imu_cov, NI * 225);
        ARENA(s.tile_win, d.tile_win, T); ARENA(s.tile_f0, d.tile_f0, T); ARENA(s.tile_n, d.tile_n, T);
        ARENA(s.lm_host, d.lm_host, L); ARENA(s.lm_optr, h->d_optr, L + B); ARENA(s.obs_raw, h->d_obs_raw, F * 3);
#undef ARENA
        size_t tot = 0;
        for (Item &q : it) { q.off = tot; tot += (q.bytes + 255) / 256 * 256; }
        void *hb = nullptr, *db = nullptr;
        HIPCHK(h, hipHostMalloc(&hb, tot ? tot : 1, hipHostMallocDefault)); h->hallocs.push_back(hb);
        HIPCHK(h, hipMalloc(&db, tot ? tot : 1)); h->allocs.push_back(db);
        HIPCHK(h, hipMemset(db, 0, tot ? tot : 1)); memset(hb, 0, tot ? tot : 1);
        auto drop = [](std::vector<void *> &v, void *p, bool host) { for (size_t i = 0; i < v.size(); i++) if (v[i] == p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); v.erase(v.begin() + i); return; } };
        for (Item &q : it) {
            drop(h->hallocs, *q.hp, true); drop(h->allocs, *q.dp, false);
            *q.hp = (char *)hb + q.off; *q.dp = (char *)db + q.off;
        }
        h->arena_h = hb; h->arena_d = db; h->arena_bytes = tot;
        h->down_a_off = (size_t)((char *)d.Ps - (char *)db); h->down_a_bytes = (size_t)((char *)d.depth - (char *)d.Ps) + sizeof(double) * L;
        // the solver's outputs: one pinned and one device block as well (a download was 24 copy commands: ~0.8 of its 1.7 ms)
        std::vector<Item> dn;
#define ARENA(hptr, dptr, cnt) dn.push_back(Item{(void **)&(hptr), (void **)&(dptr), sizeof(*(hptr)) * (size_t)(cnt), 0})
        ARENA(s.pose, d.pose, B * N * 7); ARENA(s.sb, d.sb, B * N * 9); ARENA(s.ex, d.ex, B * 7); ARENA(s.lam, d.lam, L); ARENA(s.solve_flag, d.solve_flag, L);
        ARENA(s.st, d.st, B); ARENA(h->stage.tc, d.trace_cost, B * ISV_MAX_TRACE); ARENA(h->stage.tr, d.trace_radius, B * ISV_MAX_TRACE);
        ARENA(h->stage.ts, d.trace_step, B * ISV_MAX_TRACE); ARENA(h->stage.ta, d.trace_acc, B * ISV_MAX_TRACE); ARENA(h->stage.marg, d.marg, B);
#undef ARENA
        size_t tot2 = 0;
        for (Item &q : dn) { q.off = tot2; tot2 += (q.bytes + 255) / 256 * 256; }
        void *hb2 = nullptr, *db2 = nullptr;
        HIPCHK(h, hipHostMalloc(&hb2, tot2 ? tot2 : 1, hipHostMallocDefault)); h->hallocs.push_back(hb2);
        HIPCHK(h, hipMalloc(&db2, tot2 ? tot2 : 1)); h->allocs.push_back(db2);
        HIPCHK(h, hipMemset(db2, 0, tot2 ? tot2 : 1)); memset(hb2, 0, tot2 ? tot2 : 1);
        for (Item &q : dn) {
            drop(h->hallocs, *q.hp, true); drop(h->allocs, *q.dp, false);
            *q.hp = (char *)hb2 + q.off; *q.dp = (char *)db2 + q.off;
        }
        drop(h->hallocs, h->stage.st, true); h->stage.st = s.st;      // (the solver's staging of the solve states: the same record)
        h->down_h = hb2; h->down_d = db2; h->down_bytes = tot2;
    }
    if (d.est_ex) {
        if (!d.lds_T) { h->err = "estimate_extrinsic = 1 is built for the LDS solver path only (ALL_BUF_SIZE <= 19)"; return ISV_ERR_UNSUPPORTED; }
        const size_t NPr = (size_t)c.n_frames * (c.n_frames - 1) / 2;
        TRY(dalloc(h, &d.Wex, L * 6)); TRY(dalloc(h, &d.flmx, F * 6)); TRY(dalloc(h, &d.ex_part, B * NPr * 114)); TRY(dalloc(h, &d.strip_ex, F * 12));
    }
    return ISV_OK;
}

extern "C" int isv_backend_create(const isv_config_t *cfg, isv_backend_t **out) {
    if (!cfg || !out) return ISV_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->n_frames < 3 || cfg->n_frames > ISV_MAX_FRAMES || cfg->n_vo < 2 || cfg->n_vo > cfg->n_frames - 1 ||
        cfg->max_landmarks < 0 || cfg->max_obs < 0 || cfg->max_batch < 1 || cfg->max_rollpitch < 0 || cfg->num_iterations < 0 ||
        cfg->num_iterations >= ISV_MAX_TRACE)
        return ISV_ERR_INVALID_ARG;
    if (cfg->estimate_extrinsic != 0 && cfg->estimate_extrinsic != 1) return ISV_ERR_UNSUPPORTED;     // (2 = online calibration of the initial guess: initial/, out of scope)
    isv_backend *h = new isv_backend();
    h->cfg = *cfg;
    int rc = create_impl(h);
    if (rc != ISV_OK) {
        fprintf(stderr, "isv_backend_create: %s\n", h->err.c_str());
        isv_backend_destroy(h);
        return rc;
    }
    *out = h;
    return ISV_OK;
}

static bool finite_all(const double *p, size_t n) {
    for (size_t i = 0; i < n; i++) if (!(p[i] - p[i] == 0.0)) return false;
    return true;
}

// pack one caller window into its slice of the pinned staging area (offsets fixed by the caller's first pass)
// phase (raw uploads, round 5): 0 = the whole window; 1 = what lies in the FIRST part of the upload block (states, priors, depths, IMU records), 2 = the
// rest (start frames, observation offsets and points, tiles) -- isv_batch_upload sends the first part while the host threads still pack the second
static int pack_window(isv_backend *h, int b, const isv_window_t *w, size_t L, size_t F, size_t T, std::string &err, const bool raw = false, const bool want_tiles = true, const int phase = 0) {
    const bool doA = phase != 2, doB = phase != 1;
    const isv_config_t &c = h->cfg;
    const int N = c.n_frames, Nd = h->d.N;        // real frames; device frames (+ the extrinsic's pseudo-frame)
    auto &s = h->h;
    const size_t f_off = F, lm_off = L;
    if (doA) {
    memcpy(s.Ps + (size_t)b * Nd * 3, w->Ps, sizeof(double) * N * 3); memcpy(s.Rs + (size_t)b * Nd * 9, w->Rs, sizeof(double) * N * 9);
    memcpy(s.Vs + (size_t)b * Nd * 3, w->Vs, sizeof(double) * N * 3); memcpy(s.Bas + (size_t)b * Nd * 3, w->Bas, sizeof(double) * N * 3);
    memcpy(s.Bgs + (size_t)b * Nd * 3, w->Bgs, sizeof(double) * N * 3);
    if (Nd > N) {       // pseudo-frame: "pose" = the extrinsic (k_vector2double turns it into para_Ex_Pose's twin), zero speed / biases
        memcpy(s.Ps + ((size_t)b * Nd + N) * 3, w->tic, 24); memcpy(s.Rs + ((size_t)b * Nd + N) * 9, w->ric, 72);
        memset(s.Vs + ((size_t)b * Nd + N) * 3, 0, 24); memset(s.Bas + ((size_t)b * Nd + N) * 3, 0, 24); memset(s.Bgs + ((size_t)b * Nd + N) * 3, 0, 24);
    }
    memcpy(s.tic + (size_t)b * 3, w->tic, 24); memcpy(s.ric + (size_t)b * 9, w->ric, 72);
    // every real the device will read is checked here: a NaN / Inf that reached the solve would only surface as a
    // non-finite cost many kernels later (the reference has no such check: it asserts or silently diverges)
    if (!finite_all(w->Ps, N * 3) || !finite_all(w->Rs, N * 9) || !finite_all(w->Vs, N * 3) || !finite_all(w->Bas, N * 3) ||
        !finite_all(w->Bgs, N * 3) || !finite_all(w->tic, 3) || !finite_all(w->ric, 9) ||
        (w->n_landmarks > 0 && !finite_all(w->lm_depth, w->n_landmarks)) ||
        !finite_all((const double *)w->imu, (size_t)(N - 1) * (sizeof(isv_imu_t) / sizeof(double))) ||
        !finite_all(w->pose_prior->t, 3 + 9 + 36) || !finite_all(w->vb_prior->VB, 9 + 81)) { err = "non-finite input"; return ISV_ERR_NONFINITE; }
    for (int i = 0; i < c.n_vo - 1; i++) if (!finite_all(w->relpose[i].delta_t, 3 + 9 + 36)) { err = "non-finite relative-pose prior"; return ISV_ERR_NONFINITE; }
    for (int i = 0; i < w->n_rollpitch; i++) if (!finite_all(w->rollpitch[i].R, 9 + 4)) { err = "non-finite roll/pitch prior"; return ISV_ERR_NONFINITE; }
    }
    if (doB && w->n_landmarks > 0 && !finite_all(w->obs_point, (size_t)w->n_obs * 3)) { err = "non-finite input"; return ISV_ERR_NONFINITE; }
    if (raw) {
        // (round 5) the window goes up as it is -- start frames, observation offsets, points, depths -- and k_upload_build derives the
        // solver's view (landmark / factor records, pair groups, schedule, factor stream) on the device
        const int Lw = w->n_landmarks, p0 = Lw > 0 ? w->lm_obs_ptr[0] : 0, nobs = Lw > 0 ? w->lm_obs_ptr[Lw] - p0 : 0;     // (a landmark's observations are [ptr[l], ptr[l + 1]))
        if (Lw > 0 && (size_t)(w->lm_obs_ptr[Lw - 1] - p0 - (Lw - 1)) > 65535) { err = "more than 65535 factors in one window"; return ISV_ERR_CAPACITY; }
        if (Lw > 0 && doA) memcpy(s.depth + L, w->lm_depth, sizeof(double) * Lw);
        if (doB) {
        if (Lw > 0) {
            memcpy(s.lm_host + L, w->lm_start_frame, sizeof(int32_t) * Lw);
            memcpy(s.obs_raw + (F + L) * 3, w->obs_point + (size_t)p0 * 3, sizeof(double) * 3 * (size_t)nobs);
        }
        int32_t *op = s.lm_optr + L + b;
        for (int l = 0; l <= Lw; l++) op[l] = Lw > 0 ? w->lm_obs_ptr[l] - p0 : 0;
        for (int l = 0; l < Lw; l++) s.lm_k[L + l] = op[l + 1] - op[l];        // (the tiles below; the device derives its own)
        }
        F += (size_t)(nobs - Lw); L += (size_t)Lw;
    } else
    for (int l = 0; l < w->n_landmarks; l++) {
        const int hst = w->lm_start_frame[l], o0 = w->lm_obs_ptr[l], k = w->lm_obs_ptr[l + 1] - o0;
        s.lm_host[L] = hst; s.lm_k[L] = k; s.lm_f0[L] = (int32_t)F;
        if (F - f_off > 65535) { err = "more than 65535 factors in one window"; return ISV_ERR_CAPACITY; }
        s.lm_meta[L] = (uint32_t)hst | ((uint32_t)k << 8) | ((uint32_t)(F - f_off) << 16);
        s.depth[L] = w->lm_depth[l];
        memcpy(s.lm_pts_i + L * 3, w->obs_point + (size_t)o0 * 3, 24);
        for (int o = 1; o < k; o++) {
            s.f_rec[F].lm = (int32_t)L; s.f_rec[F].ij = hst | ((hst + o) << 8);
            s.f_pts_j[F * 2] = w->obs_point[(size_t)(o0 + o) * 3]; s.f_pts_j[F * 2 + 1] = w->obs_point[(size_t)(o0 + o) * 3 + 1];
            s.f_pts_z[F] = w->obs_point[(size_t)(o0 + o) * 3 + 2];
            F++;
        }
        L++;
    }
    // tiles of <= 64 consecutive factors made of WHOLE landmarks (the linearise kernel reduces a
    // landmark's factors inside one wavefront); window_tiles() below counts them the same way
    if (want_tiles && doB) {
        size_t tf0 = f_off, tn = 0;
        for (size_t l = lm_off; l < L; l++) {
            const size_t kf = (size_t)s.lm_k[l] - 1;
            if (tn + kf > ISV_TILE) { s.tile_win[T] = b; s.tile_f0[T] = (int32_t)tf0; s.tile_n[T] = (int32_t)tn; T++; tf0 += tn; tn = 0; }
            tn += kf;
        }
        if (tn) { s.tile_win[T] = b; s.tile_f0[T] = (int32_t)tf0; s.tile_n[T] = (int32_t)tn; T++; }
    }
    // factors sorted by (host, observer) pair for the MFMA sweep: counting sort, stable in landmark order
    if (!raw) {
        const int NP = N * (N - 1) / 2;
        int32_t *off = s.pg_off + (size_t)b * (NP + 1);
        auto pidx = [N](int hh, int jj) { return hh * N - hh * (hh + 1) / 2 + (jj - hh - 1); };
        for (int p = 0; p <= NP; p++) off[p] = 0;
        for (size_t f = f_off; f < F; f++) off[pidx(s.f_rec[f].ij & 255, (s.f_rec[f].ij >> 8) & 255) + 1]++;
        for (int p = 0; p < NP; p++) off[p + 1] += off[p];
        int32_t cur[ISV_MAX_FRAMES * (ISV_MAX_FRAMES - 1) / 2];
        for (int p = 0; p < NP; p++) cur[p] = off[p];
        for (size_t f = f_off; f < F; f++) {
            const int p = pidx(s.f_rec[f].ij & 255, (s.f_rec[f].ij >> 8) & 255);
            s.pg_perm[f_off + cur[p]++] = (int32_t)(f - f_off);
        }
        // balanced schedule of the pair groups over the sweep wavefronts: longest group first onto the
        // least loaded wavefront (cost = MFMA slots: 4 per 8 factors)
        int order[ISV_MAX_FRAMES * (ISV_MAX_FRAMES - 1) / 2], wave_of[ISV_MAX_FRAMES * (ISV_MAX_FRAMES - 1) / 2];
        for (int p = 0; p < NP; p++) order[p] = p;
        std::stable_sort(order, order + NP, [&](int a2, int b2) { return off[a2 + 1] - off[a2] > off[b2 + 1] - off[b2]; });
        int load[ISV_SWEEP_WAVES] = {0}, cntw[ISV_SWEEP_WAVES] = {0};
        for (int q = 0; q < NP; q++) {
            const int p = order[q];
            int best = 0;
            for (int v = 1; v < ISV_SWEEP_WAVES; v++) if (load[v] < load[best]) best = v;
            load[best] += 1 + 4 * ((off[p + 1] - off[p] + 7) / 8);
            wave_of[p] = best; cntw[best]++;
        }
        int32_t *soff = s.pg_sched_off + (size_t)b * (ISV_SWEEP_WAVES + 1), *sched = s.pg_sched + (size_t)b * NP;
        soff[0] = 0;
        for (int v = 0; v < ISV_SWEEP_WAVES; v++) soff[v + 1] = soff[v] + cntw[v];
        int fill[ISV_SWEEP_WAVES] = {0};
        for (int hh = 0, p = 0; hh < N - 1; hh++)
            for (int jj = hh + 1; jj < N; jj++, p++) { const int v = wave_of[p]; sched[soff[v] + fill[v]++] = hh | (jj << 8) | (p << 16); }
        // the factor stream of the fused kernel: the groups back to back in schedule order, so that a wavefront walks
        // its groups in full 64-lane chunks
        int32_t *wst = s.pg_wstart + (size_t)b * (ISV_SWEEP_WAVES + 1);
        size_t q = f_off;
        for (int v = 0; v < ISV_SWEEP_WAVES; v++) {
            wst[v] = (int32_t)(q - f_off);
            for (int e = soff[v]; e < soff[v + 1]; e++) {
                const int rec = sched[e], hh = rec & 255, jj = (rec >> 8) & 255, pp = rec >> 16;
                for (int k2 = off[pp]; k2 < off[pp + 1]; k2++, q++) {
                    const size_t f = f_off + (size_t)s.pg_perm[f_off + k2];
                    s.pg_rec[2 * q] = s.f_rec[f].lm; s.pg_rec[2 * q + 1] = (int32_t)((f - f_off) | ((size_t)hh << 16) | ((size_t)jj << 24));
                    s.pg_pts[2 * q] = s.f_pts_j[2 * f]; s.pg_pts[2 * q + 1] = s.f_pts_j[2 * f + 1];
                }
            }
        }
        wst[ISV_SWEEP_WAVES] = (int32_t)(q - f_off);
    }
    if (!doA) return ISV_OK;
    if (Nd > N) {       // the IMU "factor" towards the pseudo-frame does not exist: flagged skipped (like sum_dt > 10)
        const size_t fi = (size_t)b * (Nd - 1) + (N - 1);
        memset(s.imu_in + fi * ISV_IMU_IN, 0, sizeof(double) * ISV_IMU_IN);
        s.imu_in[fi * ISV_IMU_IN + IMU_DQ + 3] = 1.0;
        for (int e = 0; e < 225; e++) s.imu_cov[fi * 225 + e] = (e % 16 == 0) ? 1.0 : 0.0;
        s.imu_skip[fi] = 1;
    }
    for (int i = 0; i < N - 1; i++) {
        const isv_imu_t &im = w->imu[i];
        double *r = s.imu_in + ((size_t)b * (Nd - 1) + i) * ISV_IMU_IN;
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
        memcpy(s.imu_cov + ((size_t)b * (Nd - 1) + i) * 225, im.covariance, sizeof(double) * 225);
        s.imu_skip[(size_t)b * (Nd - 1) + i] = im.sum_dt > 10.0;
    }
    s.se3[b] = *w->pose_prior; s.lin9[b] = *w->vb_prior;
    for (int i = 0; i < c.n_vo - 1; i++) s.relpose[(size_t)b * (c.n_vo - 1) + i] = w->relpose[i];
    for (int i = 0; i < c.max_rollpitch; i++) {
        if (i < w->n_rollpitch) s.rollpitch[(size_t)b * c.max_rollpitch + i] = w->rollpitch[i];
        else memset(&s.rollpitch[(size_t)b * c.max_rollpitch + i], 0, sizeof(isv_rollpitch_t));
    }
    s.n_rp[b] = w->n_rollpitch;
    s.margin_old[b] = w->margin_old != 0; s.header0[b] = w->header0;
    return ISV_OK;
}

// host threads for the packing pass: ISV_HOST_THREADS, else min(16, CPUs this process may run on), one per >= 32 windows
// (1024 windows: 4.6 ms on 8 threads, 3.5 ms on 16)
static int host_threads(int n) {
    int k = 0;
    if (const char *e = getenv("ISV_HOST_THREADS")) k = atoi(e);
    if (k <= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        k = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        if (k > 16) k = 16;
    }
    if (k > n / 32) k = n / 32;
    return k < 1 ? 1 : k;
}

// isv_batch_optimize starts from the state that was uploaded: twelve small device-to-device copies, as ONE kernel (a
// hipMemcpyAsync each costs ~5 us of stream time at these sizes: 60 us of every 1024-window step)
struct RestoreJobs { uint64_t *dst[12]; const uint64_t *src[12]; size_t n8[12]; };     // 8-byte words: every buffer holds doubles
__global__ __launch_bounds__(256) void k_restore(RestoreJobs j) {
    const int job = blockIdx.y;
    uint64_t *dst = j.dst[job]; const uint64_t *src = j.src[job];
    const size_t n = j.n8[job];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// (save = true: the other direction, at the end of isv_batch_upload -- the twelve hipMemcpyAsync it replaced cost ~0.2 ms of stream time)
static int restore_initial(isv_backend *h, bool save = false) {
    DevBatch &d = h->d; const isv_config_t &c = h->cfg; hipStream_t st = h->stream;
    const size_t n = d.B, N = d.N, L = d.Ltot;
    RestoreJobs j;
    int k = 0;
#define RJOB(dstp, srcp, cnt) do { static_assert(sizeof(*(srcp)) % 8 == 0, "8-byte words"); j.dst[k] = (uint64_t *)(dstp); j.src[k] = (const uint64_t *)(srcp); j.n8[k] = sizeof(*(srcp)) / 8 * (size_t)(cnt); k++; } while (0)
    RJOB(d.Ps, h->Ps0, n * N * 3); RJOB(d.Rs, h->Rs0, n * N * 9); RJOB(d.Vs, h->Vs0, n * N * 3);
    RJOB(d.Bas, h->Bas0, n * N * 3); RJOB(d.Bgs, h->Bgs0, n * N * 3); RJOB(d.depth, h->depth0, L);
    RJOB(d.tic, h->tic0, n * 3); RJOB(d.ric, h->ric0, n * 9);
    RJOB(d.se3, h->se30, n); RJOB(d.lin9, h->lin90, n); RJOB(d.relpose, h->relpose0, n * (c.n_vo - 1)); RJOB(d.rollpitch, h->rollpitch0, n * c.max_rollpitch);
#undef RJOB
    if (save) for (int q = 0; q < k; q++) { uint64_t *t_ = j.dst[q]; j.dst[q] = (uint64_t *)j.src[q]; j.src[q] = t_; }
    hipLaunchKernelGGL(k_restore, dim3(64, 12), dim3(256), 0, st, j);
    HIPCHK(h, hipGetLastError());
    return ISV_OK;
}

// pack n caller windows into the pinned staging area and copy them to the device.  Pass 1 (serial) validates the
// tracks and fixes every window's landmark / factor / tile offsets; pass 2 packs the windows on host threads.
extern "C" int isv_batch_upload(isv_backend_t *h, int32_t n, isv_window_t *const *ws) {
    if (!h || !ws || n < 1) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    const auto t_up0 = std::chrono::steady_clock::now();
    if ((size_t)n > h->capB) { h->err = "batch larger than max_batch"; return ISV_ERR_CAPACITY; }
    const isv_config_t &c = h->cfg;
    const int N = c.n_frames, Nd = h->d.N;
    auto &s = h->h;
    size_t L = 0, F = 0, T = 0, Fmax = 0, Lmax = 0;
    std::vector<size_t> t_off((size_t)n + 1);
    const bool raw = h->dev_build;
    DevBatch &d = h->d;
    hipStream_t st = h->stream;
    const size_t NIw = (size_t)(Nd - 1);
    // bytes of the windows in the upload block, and whether the block goes up in two parts (see the packing loop)
    auto used_bytes = [&]() { return (F + L) * 24 + L * 16 + (size_t)n * ((size_t)Nd * 21 * 8 + NIw * (289 * 8 + 4) + sizeof(isv_se3_prior_t) + sizeof(isv_linear9_t) +
                                      (size_t)(c.n_vo - 1) * sizeof(isv_relpose_t) + (size_t)c.max_rollpitch * sizeof(isv_rollpitch_t)); };
    bool split_copy = false;
    const size_t split_bytes = raw ? (size_t)((char *)d.tile_win - (char *)h->arena_d) : 0;
    std::atomic<int> phase_a_done{0}, copy_failed{0}, first_sent{0};
    // ONE team of host threads, three phases (round 5; the counting pass used to run on the calling thread alone: 0.7 of the 2.0 ms):
    //   1. every thread validates its windows' tracks and counts their landmarks / factors / tiles;
    //   2. thread 0 turns the counts into the windows' offsets (a prefix sum over n windows);
    //   3. every thread packs its windows into the pinned staging area.
    struct Cnt { int32_t L, F, T; int rc; };
    std::vector<Cnt> cnt((size_t)n);
    auto count_window = [&](int b) -> int {
        const isv_window_t *w = ws[b];
        if (!w || !w->Ps || !w->Rs || !w->Vs || !w->Bas || !w->Bgs || !w->tic || !w->ric || !w->imu || !w->pose_prior ||
            !w->vb_prior || !w->relpose || (w->n_rollpitch > 0 && !w->rollpitch) || w->n_landmarks < 0 ||
            (w->n_landmarks > 0 && (!w->lm_start_frame || !w->lm_obs_ptr || !w->obs_point || !w->lm_depth)))
            return ISV_ERR_INVALID_ARG;
        if (w->n_landmarks > c.max_landmarks || w->n_obs > c.max_obs || w->n_rollpitch > c.max_rollpitch) return ISV_ERR_CAPACITY;
        for (int i = 0; i < w->n_rollpitch; i++) if (w->rollpitch[i].index < 0 || w->rollpitch[i].index >= N) return ISV_ERR_INVALID_ARG;
        size_t tn = 0, Tw = 0, Fw = 0;
        for (int l = 0; l < w->n_landmarks; l++) {
            const int hst = w->lm_start_frame[l], o0 = w->lm_obs_ptr[l], k = w->lm_obs_ptr[l + 1] - o0;
            if (hst < 0 || k < 2 || hst + k > N || o0 < 0 || o0 + k > w->n_obs) return -1000;      // "bad landmark track"
            if (tn + (size_t)(k - 1) > ISV_TILE) { Tw++; tn = 0; }
            tn += (size_t)(k - 1);
            Fw += (size_t)(k - 1);
        }
        if (tn) Tw++;
        cnt[b].L = w->n_landmarks; cnt[b].F = (int32_t)Fw; cnt[b].T = (int32_t)Tw;
        return ISV_OK;
    };
    {
        const int K = host_threads(n);
        std::vector<int> rcs((size_t)K, ISV_OK);
        std::vector<std::string> errs((size_t)K);
        std::atomic<int> counted{0}, offsets_ready{0};
        auto work = [&](int k) {
            const int b0 = (int)((int64_t)n * k / K), b1 = (int)((int64_t)n * (k + 1) / K);
            for (int b = b0; b < b1 && rcs[k] == ISV_OK; b++) rcs[k] = count_window(b);
            counted.fetch_add(1, std::memory_order_release);
            if (k == 0) {
                while (counted.load(std::memory_order_acquire) < K) std::this_thread::yield();
                bool ok = true;
                for (int q = 0; q < K; q++) ok &= rcs[q] == ISV_OK;
                if (ok) {
                    for (int b = 0; b < n; b++) {
                        s.lm_off[b] = (int32_t)L; s.f_off[b] = (int32_t)F; t_off[b] = T;
                        L += (size_t)cnt[b].L; F += (size_t)cnt[b].F; T += (size_t)cnt[b].T;
                        if ((size_t)cnt[b].F > Fmax) Fmax = (size_t)cnt[b].F;
                        if ((size_t)cnt[b].L > Lmax) Lmax = (size_t)cnt[b].L;
                    }
                    t_off[n] = T; s.lm_off[n] = (int32_t)L; s.f_off[n] = (int32_t)F;
                    h->resident = 0;           // (from here on the staging area and, with the two-part copy, the device block are being overwritten)
                    split_copy = raw && K > 1 && h->arena_bytes > ((size_t)16 << 20) && used_bytes() * 10 >= h->arena_bytes * 6 && !getenv("ISV_UPLOAD_ONE_COPY");
                }
                offsets_ready.store(ok ? 1 : -1, std::memory_order_release);
            }
            int ready;
            while ((ready = offsets_ready.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
            if (ready < 0) return;
            // (round 5) a raw upload that goes up as one block is packed in TWO phases -- the first part of the block (states, priors, depths,
            // IMU records: ~45 %) and the rest (observations, tiles) -- and whichever thread finishes the first phase last sends the first
            // part while every thread packs the second: the copy (1.5 ms for 62 MB) used to start after the whole packing pass
            if (split_copy) {
                for (int b = b0; b < b1; b++) {
                    const int rc = pack_window(h, b, ws[b], (size_t)s.lm_off[b], (size_t)s.f_off[b], t_off[b], errs[k], raw, true, 1);
                    if (rc != ISV_OK) { rcs[k] = rc; break; }
                }
                if (phase_a_done.fetch_add(1, std::memory_order_acq_rel) == K - 1) {
                    bool okA = true;
                    for (int q = 0; q < K; q++) okA &= rcs[q] == ISV_OK;
                    if (okA && (hipSetDevice(h->device) != hipSuccess || hipMemcpyAsync(h->arena_d, h->arena_h, split_bytes, hipMemcpyHostToDevice, st) != hipSuccess)) copy_failed.store(1);
                    if (okA) first_sent.store(1, std::memory_order_release);
                }
                if (rcs[k] != ISV_OK) return;
                for (int b = b0; b < b1; b++) {
                    const int rc = pack_window(h, b, ws[b], (size_t)s.lm_off[b], (size_t)s.f_off[b], t_off[b], errs[k], raw, true, 2);
                    if (rc != ISV_OK) { rcs[k] = rc; return; }
                }
                return;
            }
            for (int b = b0; b < b1; b++) {
                const int rc = pack_window(h, b, ws[b], (size_t)s.lm_off[b], (size_t)s.f_off[b], t_off[b], errs[k], raw, true);
                if (rc != ISV_OK) { rcs[k] = rc; return; }
            }
        };
        if (K == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int k = 1; k < K; k++) th.emplace_back(work, k);
            work(0);
            for (auto &t : th) t.join();
        }
        for (int k = 0; k < K; k++) if (rcs[k] != ISV_OK) {
            if (rcs[k] == -1000) { h->err = "bad landmark track"; return ISV_ERR_INVALID_ARG; }
            if (rcs[k] == ISV_ERR_CAPACITY && errs[k].empty()) h->err = "window exceeds capacity"; else if (!errs[k].empty()) h->err = errs[k];
            return rcs[k];
        }
    }
    // the copies of a raw upload when the one-block copy below does not pay (a batch far below the handle's capacity): array by array
    auto enqueue_arrays = [&](int b0, int b1) -> int {
#define H2DC(dst, src, off, cnt) do { if ((cnt) > 0 && hipMemcpyAsync((dst) + (off), (src) + (off), sizeof(*(src)) * (size_t)(cnt), hipMemcpyHostToDevice, st) != hipSuccess) return ISV_ERR_DEVICE; } while (0)
        const size_t nb = (size_t)(b1 - b0), l0 = (size_t)s.lm_off[b0], l1 = (size_t)s.lm_off[b1], f0 = (size_t)s.f_off[b0], f1 = (size_t)s.f_off[b1];
        H2DC(d.Ps, s.Ps, (size_t)b0 * Nd * 3, nb * Nd * 3); H2DC(d.Rs, s.Rs, (size_t)b0 * Nd * 9, nb * Nd * 9); H2DC(d.Vs, s.Vs, (size_t)b0 * Nd * 3, nb * Nd * 3);
        H2DC(d.Bas, s.Bas, (size_t)b0 * Nd * 3, nb * Nd * 3); H2DC(d.Bgs, s.Bgs, (size_t)b0 * Nd * 3, nb * Nd * 3);
        H2DC(d.tic, s.tic, (size_t)b0 * 3, nb * 3); H2DC(d.ric, s.ric, (size_t)b0 * 9, nb * 9);
        H2DC(d.depth, s.depth, l0, l1 - l0); H2DC(d.lm_host, s.lm_host, l0, l1 - l0);
        H2DC(h->d_optr, s.lm_optr, l0 + b0, (l1 - l0) + nb); H2DC(h->d_obs_raw, s.obs_raw, (f0 + l0) * 3, ((f1 + l1) - (f0 + l0)) * 3);
        H2DC(d.imu_in, s.imu_in, (size_t)b0 * NIw * ISV_IMU_IN, nb * NIw * ISV_IMU_IN); H2DC(d.imu_cov, s.imu_cov, (size_t)b0 * NIw * 225, nb * NIw * 225);
        H2DC(d.imu_skip, s.imu_skip, (size_t)b0 * NIw, nb * NIw);
        H2DC(d.se3, s.se3, (size_t)b0, nb); H2DC(d.lin9, s.lin9, (size_t)b0, nb); H2DC(d.relpose, s.relpose, (size_t)b0 * (c.n_vo - 1), nb * (c.n_vo - 1));
        H2DC(d.rollpitch, s.rollpitch, (size_t)b0 * c.max_rollpitch, nb * c.max_rollpitch);
        H2DC(d.n_rp, s.n_rp, (size_t)b0, nb); H2DC(d.margin_old, s.margin_old, (size_t)b0, nb); H2DC(d.header0, s.header0, (size_t)b0, nb);
        // (the tiles of k_proj_linearize: the linearise API reads them whatever the solver runs; built on the host, a counting loop)
        H2DC(d.tile_win, s.tile_win, t_off[b0], t_off[b1] - t_off[b0]); H2DC(d.tile_f0, s.tile_f0, t_off[b0], t_off[b1] - t_off[b0]); H2DC(d.tile_n, s.tile_n, t_off[b0], t_off[b1] - t_off[b0]);
        H2DC(d.lm_off, s.lm_off, 0, (size_t)n + 1); H2DC(d.f_off, s.f_off, 0, (size_t)n + 1);
#undef H2DC
        return ISV_OK;
    };
    const auto t_packed = std::chrono::steady_clock::now();
    d.B = n; d.Ltot = (int32_t)L; d.Ftot = (int32_t)F; d.n_tiles = (int32_t)T;
    d.seq_hdr = nullptr;                       // (the upload path: every window of the batch is solved)
    // k_lin_gram takes one window per workgroup: right for windows of ordinary length, whatever the handle's capacity is
    // (the reference-shaped handle reserves NUM_OF_F x ALL_BUF_SIZE = 18 000 observations and sees ~2 000 factors).  A
    // batch with a very long window (BASELINE config 5: 30 000 factors in ONE window) takes the factor-parallel
    // k_proj_linearize<0> + k_sweep_mfma pair, which spreads a window over the whole GPU (6.2 against 8.3 ms per
    // optimize there).  The rule looks at the LONGEST window of the upload: a window of <= ISV_FUSED_MAX_FACTORS factors
    // gives the same bits alone and inside any batch of such windows.  (a free extrinsic only exists in k_lin_gram<true>;
    // ISV_LEGACY_VISUAL: test / measurement hook for the unfused pair)
    // k_lin_gram also stages the window's inverse depths and host points in LDS (32 B per landmark): the longest window must fit
    d.lg_lcap = (int32_t)((Lmax + 31) / 32 * 32);
    const bool lg_fits = lin_gram_lds_bytes(c.n_frames, true, c.estimate_extrinsic != 0, LG_WAVES, d.lg_lcap) <= ISV_LDS_PER_CU;
    if (c.estimate_extrinsic && !lg_fits) { h->err = "estimate_extrinsic = 1: a window has more landmarks than k_lin_gram<true> can stage in LDS"; return ISV_ERR_CAPACITY; }
    d.fused_visual = (c.estimate_extrinsic || !(h->hc.legacy_visual || Fmax > ISV_FUSED_MAX_FACTORS || !lg_fits)) ? 1 : 0;
#define H2D(dst, src, cnt) HIPCHK(h, hipMemcpyAsync(dst, src, sizeof(*(src)) * (size_t)(cnt), hipMemcpyHostToDevice, st))
    const size_t NI = (size_t)n * (Nd - 1);
    if (raw) {
        // the raw CSR goes up (69 KB per benchmark window against 110 KB of derived arrays) and the device derives the rest.  ONE copy of
        // the whole block while the batch uses most of it (or the block is small: a window or two); array by array otherwise.
        const size_t used = used_bytes();
        if (copy_failed.load()) { h->err = "isv_batch_upload: host-to-device copy failed"; return ISV_ERR_DEVICE; }
        if (first_sent.load(std::memory_order_acquire)) HIPCHK(h, hipMemcpyAsync((char *)h->arena_d + split_bytes, (char *)h->arena_h + split_bytes, h->arena_bytes - split_bytes, hipMemcpyHostToDevice, st));     // (the first part went up during the packing pass)
        else if (h->arena_bytes <= ((size_t)16 << 20) || used * 10 >= h->arena_bytes * 6) HIPCHK(h, hipMemcpyAsync(h->arena_d, h->arena_h, h->arena_bytes, hipMemcpyHostToDevice, st));
        else if (enqueue_arrays(0, n) != ISV_OK) { h->err = "isv_batch_upload: host-to-device copy failed"; return ISV_ERR_DEVICE; }
        if (isv_upload_build_enqueue(d, h->d_optr, h->d_obs_raw, c.max_landmarks > 1 ? c.max_landmarks : 1, st) != ISV_OK) { h->err = "k_upload_build launch failed"; return ISV_ERR_DEVICE; }
        if (getenv("ISV_DEBUG_UPLOAD_CHECK")) {
            // test hook: pack the same windows on the host as well and compare EVERY derived array with what the device built
            HIPCHK(h, hipStreamSynchronize(st));
            for (int b = 0; b < n; b++) { std::string e2; (void)pack_window(h, b, ws[b], (size_t)s.lm_off[b], (size_t)s.f_off[b], t_off[b], e2, false, true); }
            const size_t NPp = (size_t)N * (N - 1) / 2;
            int bad = 0;
            auto cmp = [&](const char *name, const void *host, const void *dev, size_t bytes) {
                std::vector<unsigned char> tmp(bytes ? bytes : 1);
                if (hipMemcpy(tmp.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "isv upload check: %s: copy failed\n", name); bad++; return; }
                if (memcmp(tmp.data(), host, bytes) != 0) {
                    size_t k = 0; while (k < bytes && tmp[k] == ((const unsigned char *)host)[k]) k++;
                    fprintf(stderr, "isv upload check: %s differs at byte %zu of %zu\n", name, k, bytes); bad++;
                }
            };
            cmp("lm_k", s.lm_k, d.lm_k, L * 4); cmp("lm_f0", s.lm_f0, d.lm_f0, L * 4); cmp("lm_meta", s.lm_meta, d.lm_meta, L * 4); cmp("lm_pts_i", s.lm_pts_i, d.lm_pts_i, L * 24);
            cmp("f_rec", s.f_rec, d.f_rec, F * sizeof(FactorRec)); cmp("f_pts_j", s.f_pts_j, d.f_pts_j, F * 16); cmp("f_pts_z", s.f_pts_z, d.f_pts_z, F * 8);
            cmp("pg_perm", s.pg_perm, d.pg_perm, F * 4); cmp("pg_off", s.pg_off, d.pg_off, (size_t)n * (NPp + 1) * 4); cmp("pg_sched", s.pg_sched, d.pg_sched, (size_t)n * NPp * 4);
            cmp("pg_sched_off", s.pg_sched_off, d.pg_sched_off, (size_t)n * (ISV_SWEEP_WAVES + 1) * 4); cmp("pg_wstart", s.pg_wstart, d.pg_wstart, (size_t)n * (ISV_SWEEP_WAVES + 1) * 4);
            cmp("pg_rec", s.pg_rec, d.pg_rec, F * 8); cmp("pg_pts", s.pg_pts, d.pg_pts, F * 16);
            if (bad) { h->err = "ISV_DEBUG_UPLOAD_CHECK: the device-built arrays differ from the host packer's"; return ISV_ERR_DEVICE; }
        }
    } else {
    H2D(d.lm_off, s.lm_off, n + 1); H2D(d.f_off, s.f_off, n + 1);
    H2D(d.Ps, s.Ps, (size_t)n * Nd * 3); H2D(d.Rs, s.Rs, (size_t)n * Nd * 9); H2D(d.Vs, s.Vs, (size_t)n * Nd * 3);
    H2D(d.Bas, s.Bas, (size_t)n * Nd * 3); H2D(d.Bgs, s.Bgs, (size_t)n * Nd * 3); H2D(d.tic, s.tic, (size_t)n * 3); H2D(d.ric, s.ric, (size_t)n * 9);
    H2D(d.depth, s.depth, L);
    H2D(d.lm_host, s.lm_host, L);
    H2D(d.lm_k, s.lm_k, L); H2D(d.lm_f0, s.lm_f0, L); H2D(d.lm_pts_i, s.lm_pts_i, L * 3);
    H2D(d.f_rec, s.f_rec, F); H2D(d.f_pts_j, s.f_pts_j, F * 2); H2D(d.f_pts_z, s.f_pts_z, F);
    H2D(d.lm_meta, s.lm_meta, L);
    H2D(d.tile_win, s.tile_win, T); H2D(d.tile_f0, s.tile_f0, T); H2D(d.tile_n, s.tile_n, T);
    H2D(d.pg_rec, s.pg_rec, F * 2); H2D(d.pg_pts, s.pg_pts, F * 2); H2D(d.pg_wstart, s.pg_wstart, (size_t)n * (ISV_SWEEP_WAVES + 1));
    H2D(d.pg_perm, s.pg_perm, F); H2D(d.pg_off, s.pg_off, (size_t)n * ((size_t)N * (N - 1) / 2 + 1));
    H2D(d.pg_sched, s.pg_sched, (size_t)n * ((size_t)N * (N - 1) / 2)); H2D(d.pg_sched_off, s.pg_sched_off, (size_t)n * (ISV_SWEEP_WAVES + 1));
    H2D(d.imu_in, s.imu_in, NI * ISV_IMU_IN); H2D(d.imu_cov, s.imu_cov, NI * 225); H2D(d.imu_skip, s.imu_skip, NI);
    H2D(d.se3, s.se3, n); H2D(d.lin9, s.lin9, n); H2D(d.relpose, s.relpose, (size_t)n * (c.n_vo - 1)); H2D(d.rollpitch, s.rollpitch, (size_t)n * c.max_rollpitch);
    H2D(d.n_rp, s.n_rp, n); H2D(d.margin_old, s.margin_old, n); H2D(d.header0, s.header0, n);
    }
#undef H2D
    TRY(restore_initial(h, true));              // the pristine copies isv_batch_optimize starts from
    // IMU sqrt_info once per upload (the covariances do not change during a solve)
    if (NI) hipLaunchKernelGGL(k_imu_prep, dim3((unsigned)NI), dim3(64), 0, st, d, (const int32_t *)nullptr);
    HIPCHK(h, hipGetLastError());
    const auto t_enq = std::chrono::steady_clock::now();
    HIPCHK(h, hipStreamSynchronize(st));
    if (getenv("ISV_TRACE_HANDOVER")) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "isv upload: n=%d pack %.2f ms, enqueue copies %.2f ms, wait %.2f ms\n", n, ms(t_up0, t_packed), ms(t_packed, t_enq), ms(t_enq, std::chrono::steady_clock::now()));
    }
    h->resident = n;
    return ISV_OK;
}

// the factor-linearisation kernels at the current point (pose/sb/lam); ev[1]..ev[3] bracket them
static int enqueue_linearize(isv_backend *h, bool timed) {
    DevBatch &d = h->d; hipStream_t st = h->stream;
    const size_t NI = (size_t)d.B * (d.N - 1);
    if (timed) HIPCHK(h, hipEventRecord(h->ev[1], st));
    if (d.n_tiles > 0) {
        const size_t lds = 4 * proj_lds_doubles_per_wave(d.N, 0) * sizeof(double);
        hipLaunchKernelGGL(k_proj_linearize<0>, dim3((d.n_tiles + 3) / 4), dim3(256), lds, st, d, d.pose, d.lam, d.fcost, 0);
        h->last_counts[0] += 1;
    }
    if (timed) HIPCHK(h, hipEventRecord(h->ev[2], st));
    if (NI) {
        hipLaunchKernelGGL(k_imu_raw, dim3((unsigned)(NI + 63) / 64), dim3(256), 0, st, d, d.pose, d.sb, 0);
        hipLaunchKernelGGL(k_imu_weight, dim3((unsigned)(NI + 7) / 8), dim3(256), 0, st, d, d.imu_cost, 0);
    }
    {
        hipLaunchKernelGGL(k_prior_linearize<true>, dim3(d.B), dim3(64), prior_lds_bytes(d.n_prior_slots), st, d, d.pose, d.sb, d.prior_cost, 0);
    }
    hipLaunchKernelGGL(k_cost_reduce, dim3(d.B), dim3(256), 0, st, d, d.fcost, d.imu_cost, d.prior_cost, d.cost, 0);
    if (timed) HIPCHK(h, hipEventRecord(h->ev[3], st));
    HIPCHK(h, hipGetLastError());
    return ISV_OK;
}

extern "C" int isv_batch_sync(isv_backend_t *h) {
    if (!h) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ISV_OK;
}

extern "C" int isv_batch_linearize(isv_backend_t *h, int32_t sync) {
    if (!h || !h->resident) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    DevBatch &d = h->d; hipStream_t st = h->stream;
    memset(h->last_counts, 0, sizeof(h->last_counts));
    HIPCHK(h, hipEventRecord(h->ev[0], st));
    h->prof_valid = 0;
    TRY(restore_initial(h));
    hipLaunchKernelGGL(k_vector2double, dim3(d.B), dim3(64), 0, st, d);
    TRY(enqueue_linearize(h, true));
    HIPCHK(h, hipEventRecord(h->ev[4], st));
    if (sync) HIPCHK(h, hipStreamSynchronize(st));
    return ISV_OK;
}

extern "C" int isv_batch_last_timing(isv_backend_t *h, double out_ms[8]) {
    if (!h || !out_ms) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0;
    for (int i = 0; i < 8; i++) out_ms[i] = 0;
    if (hipEventElapsedTime(&ms, h->ev[0], h->ev[4]) == hipSuccess) out_ms[0] = ms;
    if (hipEventElapsedTime(&ms, h->ev[1], h->ev[2]) == hipSuccess) out_ms[1] = ms;
    if (hipEventElapsedTime(&ms, h->ev[2], h->ev[3]) == hipSuccess) out_ms[2] = ms;
    if (h->prof_valid) {     // profiled optimize: [1] = sum k_proj_linearize<0>, [2] = sum k_sweep_mfma, [3] = sum k_rank1_mfma, [4] = sum k_build_solve*, [5] = sum k_dogleg, [6] = sum k_step_control
        for (int i = 1; i <= ISV_PROF_FAMILIES; i++) out_ms[i] = 0;
        for (int slot = 0; slot < h->cfg.num_iterations; slot++)
            for (int fam = 0; fam < ISV_PROF_FAMILIES; fam++) {
                const size_t b = ((size_t)slot * ISV_PROF_FAMILIES + fam) * 2;
                if (hipEventElapsedTime(&ms, h->prof_ev[b], h->prof_ev[b + 1]) == hipSuccess) out_ms[1 + fam] += ms;
            }
    }
    (void)hipGetLastError();      // events that were never recorded answer with an error: do not leave it for the next entry point's check
    return ISV_OK;
}

extern "C" int isv_batch_last_counts(isv_backend_t *h, int64_t out[8]) {
    if (!h || !out) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    memcpy(out, h->last_counts, sizeof(h->last_counts));
    if (h->d.lds_T && h->d.act) {          // [3] = window-iterations that were linearised and solved in the last optimize
        int32_t act[ISV_MAX_TRACE];
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemcpy(act, h->d.act, sizeof(act), hipMemcpyDeviceToHost));
        for (int i = 0; i < ISV_MAX_TRACE; i++) out[3] += act[i];
    }
    return ISV_OK;
}

// MEASUREMENT HOOK (ISV_GRAPH=1; VERDICT r2 task 5: "measure the hipGraph instead of arguing it away"): the launch chain of one
// isv_batch_optimize captured into a hipGraph and replayed while the kernel arguments (DevBatch BY VALUE: counts, offsets,
// launch variants) stay the same, i.e. while the same resident batch is solved again.  A real caller uploads new factor counts
// every frame, which changes the arguments of every node: the graph would be re-instantiated per frame (measured below in
// bench.py: the instantiation costs more than the launch gaps it removes), so this is not the default path.
static uint64_t fnv1a(const void *p, size_t n) { uint64_t x = 1469598103934665603ull; for (size_t i = 0; i < n; i++) { x ^= ((const unsigned char *)p)[i]; x *= 1099511628211ull; } return x; }

extern "C" int isv_batch_optimize(isv_backend_t *h, int32_t sync) {
    if (!h || !h->resident) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    DevBatch &d = h->d; hipStream_t st = h->stream;
    memset(h->last_counts, 0, sizeof(h->last_counts));
    const bool profile = (sync & 2) != 0 && !h->prof_ev.empty();       // per-kernel-family events only on request
    static const bool use_graph = getenv("ISV_GRAPH") != nullptr;
    if (use_graph && !profile) {
        d.sw_global = 0; d.ctl_stage_lm = 0; d.dg_stage_ph = 0;                            // (set inside the enqueue: not part of the key)
        const uint64_t key = fnv1a(&d, sizeof(d));
        HIPCHK(h, hipEventRecord(h->ev[0], st));
        if (!h->graph_exec || h->graph_key != key) {
            if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            HIPCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rcg = ISV_OK;
            if (hipMemsetAsync(d.act, 0, sizeof(int32_t) * ISV_MAX_TRACE, st) != hipSuccess) rcg = ISV_ERR_DEVICE;
            if (rcg == ISV_OK) rcg = restore_initial(h);
            if (rcg == ISV_OK) {
                hipLaunchKernelGGL(k_vector2double, dim3(d.B), dim3(64), 0, st, d);
                rcg = isv_solver_enqueue(h->d, h->hc, st, h->stream2, h->fj, h->last_counts, nullptr, h->err);
            }
            const hipError_t ec = hipStreamEndCapture(st, &g);
            if (rcg != ISV_OK || ec != hipSuccess) { if (g) (void)hipGraphDestroy(g); h->err = "hipGraph capture failed"; return rcg != ISV_OK ? rcg : ISV_ERR_DEVICE; }
            const hipError_t ei = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ei != hipSuccess) { h->graph_exec = nullptr; h->err = "hipGraphInstantiate failed"; return ISV_ERR_DEVICE; }
            h->graph_key = key;
            memcpy(h->graph_counts, h->last_counts, sizeof(h->last_counts));
        }
        memcpy(h->last_counts, h->graph_counts, sizeof(h->last_counts));
        HIPCHK(h, hipGraphLaunch(h->graph_exec, st));
        h->prof_valid = 0;
        HIPCHK(h, hipEventRecord(h->ev[4], st));
        if (sync & 1) HIPCHK(h, hipStreamSynchronize(st));
        return ISV_OK;
    }
    HIPCHK(h, hipMemsetAsync(d.act, 0, sizeof(int32_t) * ISV_MAX_TRACE, st));
    HIPCHK(h, hipEventRecord(h->ev[0], st));
    TRY(restore_initial(h));
    hipLaunchKernelGGL(k_vector2double, dim3(d.B), dim3(64), 0, st, d);
    int rc = isv_solver_enqueue(h->d, h->hc, st, h->stream2, h->fj, h->last_counts, profile ? h->prof_ev.data() : nullptr, h->err);
    h->prof_valid = profile ? 1 : 0;
    if (rc != ISV_OK) return rc;
    HIPCHK(h, hipEventRecord(h->ev[4], st));
    if (sync & 1) HIPCHK(h, hipStreamSynchronize(st));
    return ISV_OK;
}

extern "C" int isv_batch_download(isv_backend_t *h, int32_t n, isv_window_t *const *ws, isv_summary_t *summary,
                                  isv_marg_result_t *marg) {
    if (!h || !ws || n != h->resident) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    DevBatch &d = h->d; hipStream_t st = h->stream; auto &s = h->h; const isv_config_t &c = h->cfg;
    const size_t N = d.N, Nr = d.Nr, L = d.Ltot;          // device stride (incl. the extrinsic's pseudo-frame), real frames
#define D2H(dst, src, cnt) HIPCHK(h, hipMemcpyAsync(dst, src, sizeof(*(src)) * (size_t)(cnt), hipMemcpyDeviceToHost, st))
    // (round 5) a batch that uses most of the handle comes back in TWO copies (a copy command costs 20-50 us of stream time whatever its
    // size: the 24 of the array-by-array form were ~0.8 ms of a 1024-window download's 1.7); a small one array by array
    const bool two_copies = h->down_bytes > 0 && (size_t)n * 2 >= (size_t)c.max_batch && !getenv("ISV_DOWNLOAD_ARRAYS");
    if (two_copies) {
        HIPCHK(h, hipMemcpyAsync((char *)h->arena_h + h->down_a_off, (char *)h->arena_d + h->down_a_off, h->down_a_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipMemcpyAsync(h->down_h, h->down_d, h->down_bytes, hipMemcpyDeviceToHost, st));
    } else {
    D2H(s.Ps, d.Ps, n * N * 3); D2H(s.Rs, d.Rs, n * N * 9); D2H(s.Vs, d.Vs, n * N * 3); D2H(s.Bas, d.Bas, n * N * 3); D2H(s.Bgs, d.Bgs, n * N * 3);
    D2H(s.tic, d.tic, (size_t)n * 3); D2H(s.ric, d.ric, (size_t)n * 9); D2H(s.depth, d.depth, L); D2H(s.solve_flag, d.solve_flag, L);
    D2H(s.se3, d.se3, n); D2H(s.lin9, d.lin9, n); D2H(s.relpose, d.relpose, (size_t)n * (c.n_vo - 1)); D2H(s.rollpitch, d.rollpitch, (size_t)n * c.max_rollpitch);
    D2H(s.pose, d.pose, n * N * 7); D2H(s.sb, d.sb, n * N * 9); D2H(s.ex, d.ex, (size_t)n * 7); D2H(s.lam, d.lam, L);
    D2H(s.st, d.st, n);
    }
    HIPCHK(h, hipStreamSynchronize(st));
    if (!two_copies) { int rcs = isv_solver_download(h->d, st, n, h->stage, summary, marg, h->err); if (rcs != ISV_OK) return rcs; }      // (two copies: the records are here already; unpacked per window below)
    auto unpack = [&](int b) {
        isv_window_t *w = ws[b];
        memcpy(w->Ps, s.Ps + (size_t)b * N * 3, sizeof(double) * Nr * 3); memcpy(w->Rs, s.Rs + (size_t)b * N * 9, sizeof(double) * Nr * 9);
        memcpy(w->Vs, s.Vs + (size_t)b * N * 3, sizeof(double) * Nr * 3); memcpy(w->Bas, s.Bas + (size_t)b * N * 3, sizeof(double) * Nr * 3);
        memcpy(w->Bgs, s.Bgs + (size_t)b * N * 3, sizeof(double) * Nr * 3);
        memcpy(w->tic, s.tic + (size_t)b * 3, 24); memcpy(w->ric, s.ric + (size_t)b * 9, 72);
        const int l0 = s.lm_off[b];
        for (int l = 0; l < w->n_landmarks; l++) {
            w->lm_depth[l] = s.depth[l0 + l];
            if (w->lm_solve_flag) w->lm_solve_flag[l] = s.solve_flag[l0 + l];
            if (w->para_Feature) w->para_Feature[l] = s.lam[l0 + l];
        }
        *w->pose_prior = s.se3[b]; *w->vb_prior = s.lin9[b];
        for (int i = 0; i < c.n_vo - 1; i++) w->relpose[i] = s.relpose[(size_t)b * (c.n_vo - 1) + i];
        for (int i = 0; i < w->n_rollpitch; i++) w->rollpitch[i] = s.rollpitch[(size_t)b * c.max_rollpitch + i];
        if (w->para_Pose) memcpy(w->para_Pose, s.pose + (size_t)b * N * 7, sizeof(double) * Nr * 7);
        if (w->para_SpeedBias) memcpy(w->para_SpeedBias, s.sb + (size_t)b * N * 9, sizeof(double) * Nr * 9);
        if (w->para_Ex_Pose) memcpy(w->para_Ex_Pose, s.ex + (size_t)b * 7, 56);
        if (two_copies && (summary || marg)) isv_solver_unpack_window(h->stage, b, summary ? &summary[b] : nullptr, marg ? &marg[b] : nullptr);
    };
    {
        const int K = host_threads(n);
        auto work = [&](int k) { for (int b = (int)((int64_t)n * k / K), e = (int)((int64_t)n * (k + 1) / K); b < e; b++) unpack(b); };
        std::vector<std::thread> th;
        for (int k = 1; k < K; k++) th.emplace_back(work, k);
        work(0);
        for (auto &t : th) t.join();
    }
    return ISV_OK;
}

extern "C" int isv_backend_optimize_batch(isv_backend_t *h, int32_t n, isv_window_t *const *w, isv_summary_t *summary,
                                          isv_marg_result_t *marg) {
    TRY(isv_batch_upload(h, n, w));
    TRY(isv_batch_optimize(h, 1));
    return isv_batch_download(h, n, w, summary, marg);
}

extern "C" int isv_backend_optimize(isv_backend_t *h, isv_window_t *w, isv_summary_t *summary, isv_marg_result_t *marg) {
    isv_window_t *ws[1] = {w};
    return isv_backend_optimize_batch(h, 1, ws, summary, marg);
}

// Estimator::initFactorGraph (src/estimator.cpp:667-1001): solve the window without prior factors (3 x NUM_ITERATIONS
// dogleg iterations), derive the first prior factors from the solved estimate, double2vector.  The window's prior
// structs are outputs.
extern size_t init_priors_scratch_doubles(int Vo);
extern "C" int isv_backend_init_factor_graph_batch(isv_backend_t *h, int32_t n, isv_window_t *const *ws, isv_summary_t *summary, double *kld) {
    if (!h || !ws || n < 1) return ISV_ERR_INVALID_ARG;
    const isv_config_t &c = h->cfg;
    if (6 * c.n_vo + 9 > 64) { h->err = "initFactorGraph: 6 Vo + 9 > 64 is not built"; return ISV_ERR_UNSUPPORTED; }
    // the initial graph has no prior factors: zero information, identity rotations so that the residuals stay finite
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int b = 0; b < n; b++) {
        isv_window_t *w = ws[b];
        if (!w || !w->pose_prior || !w->vb_prior || !w->relpose) return ISV_ERR_INVALID_ARG;
        memset(w->pose_prior, 0, sizeof(*w->pose_prior)); memcpy(w->pose_prior->R, I3, sizeof(I3));
        memset(w->vb_prior, 0, sizeof(*w->vb_prior)); w->vb_prior->index = c.n_vo - 1;
        for (int i = 0; i < c.n_vo - 1; i++) { memset(&w->relpose[i], 0, sizeof(isv_relpose_t)); memcpy(w->relpose[i].delta_R, I3, sizeof(I3)); w->relpose[i].imu_i = i; w->relpose[i].imu_j = i + 1; }
        w->n_rollpitch = 0; w->margin_old = 0;
    }
    TRY(isv_batch_upload(h, n, ws));
    DevBatch &d = h->d; hipStream_t st = h->stream;
    const size_t per = init_priors_scratch_doubles(c.n_vo);
    if ((size_t)n > h->init_cap) {          // scratch of the one-time step: allocated on first use, kept with the handle
        if (h->init_scratch) (void)hipFree(h->init_scratch);
        if (h->init_kld) (void)hipFree(h->init_kld);
        h->init_scratch = h->init_kld = nullptr; h->init_cap = 0;
        if (hipMalloc(&h->init_scratch, (size_t)n * per * sizeof(double)) != hipSuccess || hipMalloc(&h->init_kld, (size_t)n * sizeof(double)) != hipSuccess) {
            if (h->init_scratch) (void)hipFree(h->init_scratch);
            h->init_scratch = nullptr;
            h->err = "initFactorGraph: scratch allocation failed"; return ISV_ERR_DEVICE;
        }
        h->init_cap = (size_t)n;
    }
    double *scratch = h->init_scratch, *kld_dev = h->init_kld;
    const int saved_iter = d.max_iter;
    d.max_iter = 3 * c.num_iterations < ISV_MAX_TRACE - 1 ? 3 * c.num_iterations : ISV_MAX_TRACE - 1;
    d.init_mode = 1; d.init_scratch = scratch; d.init_per_window = per; d.init_kld = kld_dev;
    memset(h->last_counts, 0, sizeof(h->last_counts));
    int rc = ISV_OK;
    if (hipMemsetAsync(d.act, 0, sizeof(int32_t) * ISV_MAX_TRACE, st) != hipSuccess) rc = ISV_ERR_DEVICE;
    if (rc == ISV_OK) {
        hipLaunchKernelGGL(k_vector2double, dim3(d.B), dim3(64), 0, st, d);
        rc = isv_solver_enqueue(h->d, h->hc, st, h->stream2, h->fj, h->last_counts, nullptr, h->err);
    }
    h->prof_valid = 0;
    d.max_iter = saved_iter; d.init_mode = 0; d.init_scratch = nullptr; d.init_kld = nullptr;
    if (rc == ISV_OK) rc = isv_batch_download(h, n, ws, summary, nullptr);
    if (rc == ISV_OK && kld && hipMemcpy(kld, kld_dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = ISV_ERR_DEVICE;
    h->resident = 0;
    return rc;
}

extern "C" int isv_backend_init_factor_graph(isv_backend_t *h, isv_window_t *w, isv_summary_t *summary, double *kld) {
    isv_window_t *ws[1] = {w};
    return isv_backend_init_factor_graph_batch(h, 1, ws, summary, kld);
}

// FeatureManager::triangulate (src/feature_tracker/feature_manager.cpp:206-258) for every landmark of the n windows whose
// estimated depth is not positive: DLT over all its views, smallest right singular vector, clamp to INIT_DEPTH.
extern "C" int isv_backend_triangulate(isv_backend_t *h, int32_t n, isv_window_t *const *ws) {
    TRY(isv_batch_upload(h, n, ws));
    DevBatch &d = h->d; hipStream_t st = h->stream; auto &s = h->h;
    if (d.Ltot) hipLaunchKernelGGL(k_triangulate, dim3((d.Ltot + 63) / 64), dim3(64), 0, st, d);
    HIPCHK(h, hipGetLastError());
    D2H(s.depth, d.depth, (size_t)d.Ltot);
    HIPCHK(h, hipStreamSynchronize(st));
    for (int b = 0; b < n; b++) {
        isv_window_t *w = ws[b];
        for (int l = 0; l < w->n_landmarks; l++) w->lm_depth[l] = s.depth[s.lm_off[b] + l];
    }
    h->resident = 0;                   // the resident copy no longer matches the caller's depths' origin; re-upload to solve
    return ISV_OK;
}

// Estimator::solveOdometry (src/estimator.cpp:461-472) for n windows with ONE hand-over: upload, triangulate on the
// device, make the triangulated depths the solve's starting point, backendOptimization, download.  The depths a caller
// would read between the two steps are overwritten by double2vector for every landmark of the window anyway.
extern "C" int isv_backend_solve_odometry_batch(isv_backend_t *h, int32_t n, isv_window_t *const *ws, isv_summary_t *summary,
                                                isv_marg_result_t *marg) {
    static const bool trace = getenv("ISV_TRACE_HANDOVER") != nullptr;       // stderr: ms of upload / triangulate + solve / download
    const auto t0 = std::chrono::steady_clock::now();
    TRY(isv_batch_upload(h, n, ws));
    const auto t1 = std::chrono::steady_clock::now();
    DevBatch &d = h->d; hipStream_t st = h->stream;
    if (d.Ltot) {
        hipLaunchKernelGGL(k_triangulate, dim3((d.Ltot + 63) / 64), dim3(64), 0, st, d);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(h->depth0, d.depth, sizeof(double) * (size_t)d.Ltot, hipMemcpyDeviceToDevice, st));        // isv_batch_optimize restores the state from the *0 copies
    }
    TRY(isv_batch_optimize(h, 1));
    const auto t2 = std::chrono::steady_clock::now();
    const int rc = isv_batch_download(h, n, ws, summary, marg);
    if (trace) {
        const auto t3 = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "isv hand-over: n=%d L=%d F=%d upload %.2f ms, triangulate + solve %.2f ms, download %.2f ms\n", n, d.Ltot, d.Ftot, ms(t0, t1), ms(t1, t2), ms(t2, t3));
    }
    return rc;
}

extern "C" int isv_backend_linearize(isv_backend_t *h, const isv_window_t *w, double *proj_strips, double *imu_strips,
                                     double *cost) {
    if (!h || !w) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    isv_window_t *ws[1] = {const_cast<isv_window_t *>(w)};
    TRY(isv_batch_upload(h, 1, ws));
    TRY(isv_batch_linearize(h, 1));
    DevBatch &d = h->d;
    // (blocking copies: the caller's buffers are pageable, and an asynchronous copy into pageable memory may still be
    // completing inside the runtime after the stream has drained)
    if (proj_strips && d.Ftot) HIPCHK(h, hipMemcpy(proj_strips, d.strip, sizeof(double) * (size_t)d.Ftot * ISV_PROJ_STRIP, hipMemcpyDeviceToHost));
    if (imu_strips) HIPCHK(h, hipMemcpy(imu_strips, d.imu_strip, sizeof(double) * (size_t)(d.Nr - 1) * ISV_IMU_STRIP, hipMemcpyDeviceToHost));
    if (cost) HIPCHK(h, hipMemcpy(cost, d.cost, sizeof(double), hipMemcpyDeviceToHost));
    return ISV_OK;
}

// ---- per-window result records in DEVICE memory (the one exchange step of the multi-GPU configuration is an
// all-gather of these records; the caller owns the buffer, e.g. a torch tensor handed to RCCL) ------------------------
// record = [para_Pose 7N | para_SpeedBias 9N | inverse depths, zero padded to max_landmarks |
//           final_cost, initial_cost, iterations, termination, num_successful, radius, header0, n_landmarks]
__global__ void k_pack_results(DevBatch d, double *dst, int64_t rec, int maxL) {
    const int w = blockIdx.x, t = threadIdx.x, N = d.Nr, Nd = d.N;
    double *o = dst + (size_t)w * rec;
    for (int i = t; i < 7 * N; i += blockDim.x) o[i] = d.pose[(size_t)w * Nd * 7 + i];
    for (int i = t; i < 9 * N; i += blockDim.x) o[7 * N + i] = d.sb[(size_t)w * Nd * 9 + i];
    const int l0 = d.lm_off[w], Lw = d.lm_off[w + 1] - l0;
    for (int i = t; i < maxL; i += blockDim.x) o[16 * N + i] = i < Lw ? d.lam[l0 + i] : 0.0;
    if (t == 0) {
        const SolveState &st = d.st[w];
        double *q = o + 16 * N + maxL;
        q[0] = st.x_cost; q[1] = st.initial_cost; q[2] = st.iteration; q[3] = st.termination; q[4] = st.num_successful;
        q[5] = st.radius; q[6] = d.header0[w]; q[7] = Lw;
    }
}

extern "C" int64_t isv_result_record_doubles(const isv_backend_t *h) {
    return h ? 16 * (int64_t)h->cfg.n_frames + h->cfg.max_landmarks + 8 : 0;
}

extern "C" int isv_batch_pack_results(isv_backend_t *h, void *device_dst, void *stream) {
    if (!h || !device_dst || !h->resident) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    DevBatch &d = h->d;
    // (NULL is the legacy default stream, a caller stream like any other: only the sentinel selects the handle's own)
    hipStream_t dst_stream = stream == ISV_STREAM_OF_HANDLE ? h->stream : (hipStream_t)stream;
    if (dst_stream != h->stream) {           // order the pack after everything enqueued on the handle's stream, without a host sync
        HIPCHK(h, hipEventRecord(h->pk[0], h->stream));
        HIPCHK(h, hipStreamWaitEvent(dst_stream, h->pk[0], 0));
    }
    hipLaunchKernelGGL(k_pack_results, dim3(d.B), dim3(256), 0, dst_stream, d, (double *)device_dst, isv_result_record_doubles(h), h->cfg.max_landmarks);
    HIPCHK(h, hipGetLastError());
    if (dst_stream != h->stream) {           // ... and the handle's next launch (which overwrites the states) after the pack
        HIPCHK(h, hipEventRecord(h->pk[1], dst_stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->pk[1], 0));
    }
    return ISV_OK;
}

// test hook: prior strips / per-block costs of window 0 after isv_backend_linearize
extern "C" int isv_debug_read(isv_backend_t *h, int32_t what, double *out, int64_t count) {
    if (!h || !out) return ISV_ERR_INVALID_ARG;
    ENTER(h);
    DevBatch &d = h->d; hipStream_t st = h->stream;
    const double *src = nullptr;
    switch (what) {
    case 0: src = d.prior_strip; break;
    case 1: src = d.prior_cost; break;
    case 2: src = d.imu_sqrt; break;
    case 3: src = d.fcost; break;
    case 4: src = d.imu_cost; break;
    case 5: src = d.pose; break;
    case 6: src = d.lam; break;
    case 7: src = d.cost; break;
    case 22: src = d.strip_ex; if (!src) return ISV_ERR_INVALID_ARG; break;
    default: return isv_solver_debug_read(h->d, st, what, out, count, h->err);
    }
    D2H(out, src, count);
    HIPCHK(h, hipStreamSynchronize(st));
    return ISV_OK;
}
