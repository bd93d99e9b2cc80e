// Memory-safety run of the window manager's host logic (is-vins_amd/csrc/isv_estimator.cpp) under
// AddressSanitizer + UBSan on the CPU: a stand-in solver behind isv_solver_vtbl_t (test seam) that performs no
// optimisation -- it gives every landmark a depth, flags a few as failures and hands back fixed marginalisation
// factors -- so that thousands of processIMU / processImage / slideWindow steps with tracks appearing, ageing and
// dying run through both slideWindow branches, removeBackShiftDepth, removeFront and removeFailures.
// Built and run by tests/test_native_sanitize.py; not part of the library.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/isvins_estimator.h"

// the HIP backend is not linked into this binary: the estimator only reaches it through isv_estimator_create
extern "C" int isv_backend_create(const isv_config_t *, isv_backend_t **) { return ISV_ERR_DEVICE; }
extern "C" void isv_backend_destroy(isv_backend_t *) {}
extern "C" int isv_backend_triangulate(isv_backend_t *, int32_t, isv_window_t *const *) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_init_factor_graph(isv_backend_t *, isv_window_t *, isv_summary_t *, double *) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_optimize_batch(isv_backend_t *, int32_t, isv_window_t *const *, isv_summary_t *, isv_marg_result_t *) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_init_factor_graph_batch(isv_backend_t *, int32_t, isv_window_t *const *, isv_summary_t *, double *) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_solve_odometry_batch(isv_backend_t *, int32_t, isv_window_t *const *, isv_summary_t *, isv_marg_result_t *) { return ISV_ERR_DEVICE; }
// (the device-resident mode is only reachable with the HIP backend: its entry points are stubs here, like the rest)
extern "C" const char *isv_backend_last_error(const isv_backend_t *) { return "no device in the sanitizer harness"; }
extern "C" int isv_backend_seq_enable(isv_backend_t *, int32_t) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_seq_seed(isv_backend_t *, int32_t, isv_window_t *const *, const int32_t *, const isv_seq_track_t *const *, const double *const *) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_seq_frame(isv_backend_t *, int32_t, const isv_seq_frame_t *, isv_seq_result_t *, int32_t *const *, isv_marg_result_t *) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_seq_download(isv_backend_t *, int32_t, isv_window_t *, int32_t, double *, int32_t *) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_seq_flush(isv_backend_t *, int32_t, const int32_t *, const int32_t *) { return ISV_ERR_DEVICE; }
extern "C" int isv_backend_seq_marg(isv_backend_t *, int32_t, isv_marg_result_t *) { return ISV_ERR_DEVICE; }

static int g_calls = 0;
static int st_triangulate(void *, int32_t n, isv_window_t *const *ws) {
    for (int b = 0; b < n; b++) for (int l = 0; l < ws[b]->n_landmarks; l++) if (!(ws[b]->lm_depth[l] > 0)) ws[b]->lm_depth[l] = 4.0;
    return ISV_OK;
}
static int st_init(void *, isv_window_t *w, isv_summary_t *s, double *kld) {
    memset(s, 0, sizeof(*s)); s->iterations = 3; if (kld) *kld = 0;
    for (int l = 0; l < w->n_landmarks; l++) w->lm_solve_flag[l] = 1;
    w->n_rollpitch = 0;
    return ISV_OK;
}
static int st_optimize(void *, int32_t n, isv_window_t *const *ws, isv_summary_t *s, isv_marg_result_t *m) {
    for (int b = 0; b < n; b++) {
        isv_window_t *w = ws[b];
        memset(&s[b], 0, sizeof(s[b])); s[b].iterations = 2;
        memset(&m[b], 0, sizeof(m[b]));
        for (int l = 0; l < w->n_landmarks; l++) { w->lm_solve_flag[l] = ((g_calls + l) % 17 == 0) ? 2 : 1; w->lm_depth[l] = 3.0 + 0.01 * l; }
        if (w->margin_old) {
            m[b].valid = 1;
            static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            memcpy(m[b].forward_pose_prior.R, I3, sizeof(I3)); memcpy(m[b].backward_relpose.delta_R, I3, sizeof(I3)); memcpy(m[b].backward_rollpitch.R, I3, sizeof(I3));
        }
        g_calls++;
    }
    return ISV_OK;
}
static int st_solve_odometry(void *c, int32_t n, isv_window_t *const *ws, isv_summary_t *s, isv_marg_result_t *m) {
    const int rc = st_triangulate(c, n, ws);
    return rc != ISV_OK ? rc : st_optimize(c, n, ws, s, m);
}

int main() {
    const int N = 7, Nvo = 3, S = 19, FRAMES = 120;
    isv_estimator_params_t p;
    memset(&p, 0, sizeof(p));
    p.cfg.n_frames = N; p.cfg.n_vo = Nvo; p.cfg.max_landmarks = 200; p.cfg.max_obs = 200 * N; p.cfg.max_rollpitch = Nvo + 1;
    p.cfg.max_batch = S; p.cfg.num_iterations = 10; p.cfg.gravity[2] = 9.81; p.cfg.init_depth = 5.0; p.cfg.alpha = 0.1;
    p.cfg.proj_sqrt_info[0] = p.cfg.proj_sqrt_info[3] = 460.0;
    p.ric[0] = p.ric[4] = p.ric[8] = 1.0;
    p.acc_n = 0.2; p.gyr_n = 0.004; p.acc_w = 0.001; p.gyr_w = 0.0001; p.min_parallax = 10.0 / 460.0;
    isv_solver_vtbl_t vt = {nullptr, st_triangulate, st_init, st_optimize, nullptr, st_solve_odometry};
    isv_estimator_t *e = nullptr;
    if (isv_estimator_create(&p, S, &e) == ISV_OK) { fprintf(stderr, "create without a backend must fail\n"); return 1; }
    if (isv_estimator_create_with_solver(&p, S, &vt, &e) != ISV_OK) { fprintf(stderr, "create_with_solver failed\n"); return 1; }
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> U(-0.5, 0.5);
    std::vector<double> bootP(N * 3, 0.0), bootR(N * 9, 0.0), bootV(N * 3, 0.0);
    for (int i = 0; i < N; i++) { bootR[i * 9] = bootR[i * 9 + 4] = bootR[i * 9 + 8] = 1.0; bootP[i * 3] = 0.1 * i; }
    long solved = 0;
    std::vector<double> off(FRAMES, 0.0);            // camera motion: stands still on frames 4..6 of every 9 (low parallax -> MARGIN_SECOND_NEW)
    for (int fr = 1; fr < FRAMES; fr++) off[fr] = off[fr - 1] + ((fr % 9 >= 4 && fr % 9 <= 6) ? 0.0 : 0.03);
    int old_steps = 0, new_steps = 0;
    for (int fr = 0; fr < FRAMES; fr++) {
        for (int s = 0; s < S; s++) {
            for (int k = 0; k < 5; k++) {
                const double acc[3] = {U(rng), U(rng), 9.81 + U(rng)}, gyr[3] = {0.1 * U(rng), 0.1 * U(rng), 0.1 * U(rng)};
                if (isv_estimator_process_imu(e, s, 0.02, acc, gyr) != ISV_OK) return 1;
            }
            // a moving band of feature ids: tracks are born, live a few frames and die; some frames carry < 20 tracked ids
            std::vector<int32_t> ids; std::vector<double> pts;
            const int base = fr * 6 + s, count = (fr % 13 == 5) ? 12 : 60;
            for (int k = 0; k < count; k++) {
                const int id = base + k;
                ids.push_back(id);
                pts.push_back(0.01 * (id % 50) + off[fr]); pts.push_back(0.02 * (id % 31)); pts.push_back(1.0);
            }
            int32_t st[8];
            isv_estimator_status(e, s, st);
            if (st[0] == 0 && st[1] == N - 1 && isv_estimator_set_bootstrap(e, s, bootP.data(), bootR.data(), bootV.data()) != ISV_OK) return 1;
            if (isv_estimator_push_image(e, s, 0.1 * fr, (int32_t)ids.size(), ids.data(), pts.data()) != ISV_OK) return 1;
        }
        const int n = isv_estimator_step(e);
        if (n < 0) { fprintf(stderr, "step failed: %d %s\n", n, isv_estimator_last_error(e)); return 1; }
        solved += n;
        if (n > 0) { int32_t st[8]; isv_estimator_status(e, 0, st); (st[2] ? old_steps : new_steps)++; }
    }
    if (old_steps == 0 || new_steps == 0) { fprintf(stderr, "only one slideWindow branch was taken (%d / %d)\n", old_steps, new_steps); return 1; }
    int old_seen = 0, new_seen = 0;
    for (int s = 0; s < S; s++) {
        int32_t st[8];
        isv_estimator_status(e, s, st);
        if (st[0] != 1 || st[1] != N - 1) { fprintf(stderr, "sequence %d did not reach NON_LINEAR\n", s); return 1; }
        (st[2] ? old_seen : new_seen)++;
        std::vector<double> rows((size_t)FRAMES * 13);
        const int r = isv_estimator_trajectory(e, s, 1, rows.data(), FRAMES);
        if (r != FRAMES - (N - 1)) { fprintf(stderr, "sequence %d: %d rows\n", s, r); return 1; }
    }
    isv_estimator_destroy(e);
    printf("ok: %ld sequence-frames solved; steps with MARGIN_OLD %d / MARGIN_SECOND_NEW %d\n", solved, old_steps, new_steps);
    return 0;
}
