// isv_replay -- headless replay of recorded camera-IMU streams through the native window manager on the MI355X
// (include/isvins_estimator.h): what `run_euroc` does after the feature tracker in the reference (System::ProcessBackEnd
// feeding Estimator::processIMU / processImage, src/System.cpp:246-413), without OpenCV / Pangolin, for many sequences.
//
//   isv_replay STREAM [--sequences S] [--groups K] [--out DIR] [--write W]
//
// STREAM is a text file (tests/sequence_harness.py::write_stream writes one from the simulator; a feature tracker's
// output can be dumped in the same form):
//   config N Nvo max_landmarks num_iterations pixel_sqrt_info g alpha init_depth acc_n gyr_n acc_w gyr_w min_parallax
//   ric r00 .. r22          tic x y z
//   imu dt ax ay az gx gy gz                       one Estimator::processIMU call
//   boot                                            followed by N lines  px py pz  r00..r22  vx vy vz  (replaces
//                                                   initialStructure; given before the frame that fills the window)
//   frame stamp n                                   followed by n lines  id x y z   (Estimator::processImage)
// The stream is replayed into S sequences (copies), split over K groups; every group has its own estimator (its own
// backend handle and streams) and host thread, so one group's packing overlaps another group's solve.  Writes
// DIR/pose_output_<s>.txt (src/System.cpp:401-410 row format) for the first W sequences and prints one JSON line.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../include/isvins_estimator.h"

namespace {

struct Event {
    int kind;                       // 0 imu, 1 boot, 2 frame
    double dt = 0, acc[3] = {0, 0, 0}, gyr[3] = {0, 0, 0};
    std::vector<double> P, R, V;    // boot
    double stamp = 0;
    std::vector<int32_t> ids;       // frame
    std::vector<double> pts;
};

struct Stream {
    isv_estimator_params_t params{};
    std::vector<Event> events;
    int n_frames = 0;
};

bool read_stream(const char *path, Stream &s, std::string &err) {
    FILE *f = fopen(path, "r");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    char tag[32];
    bool have_cfg = false;
    while (fscanf(f, "%31s", tag) == 1) {
        if (tag[0] == '#') { int c; while ((c = fgetc(f)) != '\n' && c != EOF) {} continue; }
        if (!strcmp(tag, "config")) {
            isv_config_t &c = s.params.cfg;
            double psi, g;
            if (fscanf(f, "%d %d %d %d %lf %lf %lf %lf %lf %lf %lf %lf %lf", &c.n_frames, &c.n_vo, &c.max_landmarks, &c.num_iterations, &psi, &g,
                       &c.alpha, &c.init_depth, &s.params.acc_n, &s.params.gyr_n, &s.params.acc_w, &s.params.gyr_w, &s.params.min_parallax) != 13) { err = "bad config line"; fclose(f); return false; }
            c.max_obs = c.max_landmarks * c.n_frames; c.max_rollpitch = c.n_vo + 1; c.max_batch = 1; c.estimate_extrinsic = 0;
            c.proj_sqrt_info[0] = c.proj_sqrt_info[3] = psi; c.proj_sqrt_info[1] = c.proj_sqrt_info[2] = 0;
            c.gravity[0] = c.gravity[1] = 0; c.gravity[2] = g;
            have_cfg = true;
        } else if (!strcmp(tag, "ric")) {
            for (int k = 0; k < 9; k++) if (fscanf(f, "%lf", &s.params.ric[k]) != 1) { err = "bad ric"; fclose(f); return false; }
        } else if (!strcmp(tag, "tic")) {
            for (int k = 0; k < 3; k++) if (fscanf(f, "%lf", &s.params.tic[k]) != 1) { err = "bad tic"; fclose(f); return false; }
        } else if (!strcmp(tag, "imu")) {
            Event e; e.kind = 0;
            if (fscanf(f, "%lf %lf %lf %lf %lf %lf %lf", &e.dt, &e.acc[0], &e.acc[1], &e.acc[2], &e.gyr[0], &e.gyr[1], &e.gyr[2]) != 7) { err = "bad imu line"; fclose(f); return false; }
            s.events.push_back(std::move(e));
        } else if (!strcmp(tag, "boot")) {
            if (!have_cfg) { err = "boot before config"; fclose(f); return false; }
            Event e; e.kind = 1;
            const int N = s.params.cfg.n_frames;
            e.P.resize(N * 3); e.R.resize(N * 9); e.V.resize(N * 3);
            for (int i = 0; i < N; i++) {
                for (int k = 0; k < 3; k++) if (fscanf(f, "%lf", &e.P[i * 3 + k]) != 1) { err = "bad boot line"; fclose(f); return false; }
                for (int k = 0; k < 9; k++) if (fscanf(f, "%lf", &e.R[i * 9 + k]) != 1) { err = "bad boot line"; fclose(f); return false; }
                for (int k = 0; k < 3; k++) if (fscanf(f, "%lf", &e.V[i * 3 + k]) != 1) { err = "bad boot line"; fclose(f); return false; }
            }
            s.events.push_back(std::move(e));
        } else if (!strcmp(tag, "frame")) {
            Event e; e.kind = 2;
            int n = 0;
            if (fscanf(f, "%lf %d", &e.stamp, &n) != 2 || n < 0) { err = "bad frame line"; fclose(f); return false; }
            e.ids.resize(n); e.pts.resize((size_t)n * 3);
            for (int i = 0; i < n; i++)
                if (fscanf(f, "%d %lf %lf %lf", &e.ids[i], &e.pts[i * 3], &e.pts[i * 3 + 1], &e.pts[i * 3 + 2]) != 4) { err = "bad feature line"; fclose(f); return false; }
            s.events.push_back(std::move(e));
            s.n_frames++;
        } else { err = std::string("unknown record '") + tag + "'"; fclose(f); return false; }
    }
    fclose(f);
    if (!have_cfg) { err = "no config line"; return false; }
    return true;
}

struct GroupResult {
    int rc = 0;
    std::string err;
    long solved = 0;                // sequence-frames solved in steady state
    double seconds = 0;             // wall time of the steady-state part
    double step_ms_sum[6] = {0, 0, 0, 0, 0, 0};
    long steps = 0;
};

void run_group(const Stream &st, int first_seq, int n_seq, const std::string &out_dir, int write_upto, int feed_threads, GroupResult &res) {
    isv_estimator_t *e = nullptr;
    res.rc = isv_estimator_create(&st.params, n_seq, &e);
    if (res.rc != ISV_OK) { res.err = "isv_estimator_create failed (a GPU is required)"; return; }
    const int N = st.params.cfg.n_frames;
    int frames_seen = 0;
    using clk = std::chrono::steady_clock;
    clk::time_point t_start;
    bool timing = false;
    // host threads of this group for the per-sequence feed (the sequences are independent; processIMU / push_image on
    // different sequences may run concurrently)
    int T = feed_threads;
    if (T > n_seq / 8) T = n_seq / 8;
    if (T < 1) T = 1;
    auto for_sequences = [&](auto body) -> int {
        std::vector<int> rcs(T, ISV_OK);
        auto work = [&](int k) { for (int s = (int)((long)n_seq * k / T), end = (int)((long)n_seq * (k + 1) / T); s < end && rcs[k] == ISV_OK; s++) rcs[k] = body(s); };
        std::vector<std::thread> th;
        for (int k = 1; k < T; k++) th.emplace_back(work, k);
        work(0);
        for (auto &t : th) t.join();
        for (int k = 0; k < T; k++) if (rcs[k] != ISV_OK) return rcs[k];
        return (int)ISV_OK;
    };
    std::vector<double> dts, accs, gyrs;            // the IMU samples since the last frame
    const Event *boot = nullptr;
    for (const Event &ev : st.events) {
        int rc = ISV_OK;
        if (ev.kind == 0) {
            dts.push_back(ev.dt); accs.insert(accs.end(), ev.acc, ev.acc + 3); gyrs.insert(gyrs.end(), ev.gyr, ev.gyr + 3);
            continue;
        } else if (ev.kind == 1) {
            boot = &ev;
            continue;
        } else {
            rc = for_sequences([&](int s) {
                int r = isv_estimator_process_imu_n(e, s, (int32_t)dts.size(), dts.data(), accs.data(), gyrs.data());
                if (r == ISV_OK && boot) r = isv_estimator_set_bootstrap(e, s, boot->P.data(), boot->R.data(), boot->V.data());
                if (r == ISV_OK) r = isv_estimator_push_image(e, s, ev.stamp, (int32_t)ev.ids.size(), ev.ids.data(), ev.pts.data());
                return r;
            });
            dts.clear(); accs.clear(); gyrs.clear(); boot = nullptr;
            if (rc == ISV_OK) {
                const int n = isv_estimator_step(e);
                if (n < 0) rc = n;
                else if (timing) {
                    double ms[6];
                    isv_estimator_last_step_ms(e, ms);
                    for (int k = 0; k < 6; k++) res.step_ms_sum[k] += ms[k];
                    res.steps++; res.solved += n;
                }
            }
            frames_seen++;
            if (frames_seen == N + 2 && !timing) { timing = true; t_start = clk::now(); }      // steady state: past the first solves (initFactorGraph)
        }
        if (rc != ISV_OK) { res.rc = rc; res.err = isv_estimator_last_error(e); isv_estimator_destroy(e); return; }
    }
    if (timing) res.seconds = std::chrono::duration<double>(clk::now() - t_start).count();
    for (int s = 0; s < n_seq; s++) {
        const int gs = first_seq + s;
        if (gs >= write_upto || out_dir.empty()) continue;
        const int rows = isv_estimator_trajectory(e, s, 0, nullptr, 0);
        std::vector<double> buf((size_t)(rows > 0 ? rows : 1) * 8);
        isv_estimator_trajectory(e, s, 0, buf.data(), rows);
        const std::string path = out_dir + "/pose_output_" + std::to_string(gs) + ".txt";
        FILE *f = fopen(path.c_str(), "w");
        if (!f) { res.rc = ISV_ERR_INVALID_ARG; res.err = "cannot write " + path; break; }
        for (int r = 0; r < rows; r++) {            // ofs << fixed << stamp << " " << p << " " << q.w() q.x() q.y() q.z()   src/System.cpp:408-409
            const double *x = &buf[(size_t)r * 8];
            fprintf(f, "%.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
        }
        fclose(f);
    }
    isv_estimator_destroy(e);
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: isv_replay STREAM [--sequences S] [--groups K] [--out DIR] [--write W]\n"); return 2; }
    int S = 1, K = 1, W = 1;
    std::string out_dir;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--sequences") && i + 1 < argc) S = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--groups") && i + 1 < argc) K = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_dir = argv[++i];
        else if (!strcmp(argv[i], "--write") && i + 1 < argc) W = atoi(argv[++i]);
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (S < 1 || K < 1 || K > S) { fprintf(stderr, "need 1 <= groups <= sequences\n"); return 2; }
    Stream st;
    std::string err;
    if (!read_stream(argv[1], st, err)) { fprintf(stderr, "isv_replay: %s\n", err.c_str()); return 1; }
    int feed_threads = (int)std::thread::hardware_concurrency() / K;      // per group, for the per-sequence feed
    if (feed_threads > 8) feed_threads = 8;
    if (feed_threads < 1) feed_threads = 1;
    std::vector<GroupResult> res(K);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; k++) {
        const int a = (int)((long)S * k / K), b = (int)((long)S * (k + 1) / K);
        th.emplace_back(run_group, std::cref(st), a, b - a, out_dir, W, feed_threads, std::ref(res[k]));
    }
    for (auto &t : th) t.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long solved = 0, steps = 0;
    double secs = 0, ms[6] = {0, 0, 0, 0, 0, 0};
    for (const GroupResult &r : res) {
        if (r.rc != ISV_OK) { fprintf(stderr, "isv_replay: status %d: %s\n", r.rc, r.err.c_str()); return 1; }
        solved += r.solved; steps += r.steps;
        if (r.seconds > secs) secs = r.seconds;
        for (int k = 0; k < 6; k++) ms[k] += r.step_ms_sum[k];
    }
    printf("{\"sequences\": %d, \"groups\": %d, \"frames_in_stream\": %d, \"n_frames_window\": %d, \"solved_in_steady_state\": %ld, "
           "\"steady_state_seconds\": %.6f, \"frames_per_second\": %.1f, \"wall_seconds\": %.3f, "
           "\"mean_step_ms\": {\"step\": %.3f, \"features_and_packing\": %.3f, \"triangulate\": %.3f, \"init_factor_graph\": %.3f, \"solve_odometry\": %.3f, \"readback_and_slide\": %.3f}}\n",
           S, K, st.n_frames, st.params.cfg.n_frames, solved, secs, secs > 0 ? solved / secs : 0.0, wall,
           steps ? ms[0] / steps : 0.0, steps ? ms[1] / steps : 0.0, steps ? ms[2] / steps : 0.0, steps ? ms[3] / steps : 0.0, steps ? ms[4] / steps : 0.0,
           steps ? ms[5] / steps : 0.0);
    return 0;
}
