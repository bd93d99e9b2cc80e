// isv_replay -- headless replay of recorded camera-IMU streams through the native window manager on the MI355X
// (include/isvins_estimator.h): what `run_euroc` does after the feature tracker in the reference (System::ProcessBackEnd
// feeding Estimator::processIMU / processImage, src/System.cpp:246-413), without OpenCV / Pangolin, for many sequences.
//
//   isv_replay STREAM [--sequences S] [--groups K] [--out DIR] [--write W]
//   isv_replay --euroc MAV0_DIR --tracks TRACKS.csv --config CONFIG.txt [--sequences S] ...
//   ... --dump-events FILE   write the processIMU / processImage calls the input turns into and stop (no GPU needed)
//
// --euroc reads the dataset files the reference's test/run_euroc.cpp reads, in its formats: MAV0_DIR/imu0/data.csv
// (`timestamp[ns],w_x,w_y,w_z,a_x,a_y,a_z`, test/run_euroc.cpp:26-50) and, in place of cam0 images + the feature
// tracker (OpenCV, out of scope), a feature-track table TRACKS.csv (`timestamp[ns],id,x,y,z` per observation, frames in
// time order).  IMU samples and frames are paired the way System::getMeasurements / ProcessBackEnd do
// (src/System.cpp:160-202, 262-296): every sample before the image time, the first one at or after it re-used by the next
// frame, and a linear interpolation at the image time when a sample falls behind it.  CONFIG.txt holds the `config` /
// `ric` / `tic` lines below and, optionally, the `boot` rows for the frame that fills the window.
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

// ---- EuRoC-format input (test/run_euroc.cpp readers + the measurement pairing of System) -------------------------
struct ImuSample { double t; double gyr[3], acc[3]; };
struct Frame { double t; std::vector<int32_t> ids; std::vector<double> pts; };

bool read_imu_csv(const std::string &path, std::vector<ImuSample> &out, std::string &err) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) { err = "cannot open " + path; return false; }
    char line[512];
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#' || line[0] == '\n' || line[0] == '\r') continue;
        ImuSample s; double ns;
        if (sscanf(line, "%lf,%lf,%lf,%lf,%lf,%lf,%lf", &ns, &s.gyr[0], &s.gyr[1], &s.gyr[2], &s.acc[0], &s.acc[1], &s.acc[2]) != 7) { err = "bad line in " + path; fclose(f); return false; }
        s.t = ns / 1e9;                                    // dStampNSec / 1e9  (test/run_euroc.cpp:47)
        out.push_back(s);
    }
    fclose(f);
    return true;
}

bool read_tracks_csv(const std::string &path, std::vector<Frame> &out, std::string &err) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) { err = "cannot open " + path; return false; }
    char line[512];
    double last_ns = -1;
    while (fgets(line, sizeof(line), f)) {
        if (line[0] == '#' || line[0] == '\n' || line[0] == '\r') continue;
        double ns, x, y, z; int id;
        if (sscanf(line, "%lf,%d,%lf,%lf,%lf", &ns, &id, &x, &y, &z) != 5) { err = "bad line in " + path; fclose(f); return false; }
        if (out.empty() || ns != last_ns) { Frame fr; fr.t = ns / 1e9; out.push_back(fr); last_ns = ns; }
        out.back().ids.push_back(id);
        out.back().pts.push_back(x); out.back().pts.push_back(y); out.back().pts.push_back(z);
    }
    fclose(f);
    return true;
}

// System::getMeasurements + the IMU loop of System::ProcessBackEnd (td = 0), offline: the events Estimator::processIMU /
// processImage receive for these IMU samples and frames
void assemble_events(const std::vector<ImuSample> &imu, const std::vector<Frame> &frames, const Event *boot, int N, Stream &st) {
    size_t q = 0;                                  // imu_buf.front()
    double current_time = -1, last[6] = {0, 0, 0, 0, 0, 0};
    int frames_out = 0;
    for (const Frame &fr : frames) {
        if (q >= imu.size() || !(imu.back().t > fr.t)) break;          // "wait for imu": offline, the end of the data
        if (!(imu[q].t < fr.t)) continue;                                // "throw img, only should happen at the beginning"
        std::vector<size_t> take;
        while (imu[q].t < fr.t) take.push_back(q++);
        take.push_back(q);                                               // the first sample at or after the image stays queued
        for (size_t k : take) {
            const ImuSample &m = imu[k];
            Event e; e.kind = 0;
            if (m.t <= fr.t) {
                if (current_time < 0) current_time = m.t;
                e.dt = m.t - current_time;
                current_time = m.t;
                for (int a = 0; a < 3; a++) { last[a] = m.acc[a]; last[3 + a] = m.gyr[a]; }
            } else {                                                     // interpolate at the image time
                const double dt_1 = fr.t - current_time, dt_2 = m.t - fr.t;
                current_time = fr.t;
                const double w1 = dt_2 / (dt_1 + dt_2), w2 = dt_1 / (dt_1 + dt_2);
                for (int a = 0; a < 3; a++) { last[a] = w1 * last[a] + w2 * m.acc[a]; last[3 + a] = w1 * last[3 + a] + w2 * m.gyr[a]; }
                e.dt = dt_1;
            }
            for (int a = 0; a < 3; a++) { e.acc[a] = last[a]; e.gyr[a] = last[3 + a]; }
            st.events.push_back(e);
        }
        if (boot && frames_out == N - 1) st.events.push_back(*boot);
        Event e; e.kind = 2; e.stamp = fr.t; e.ids = fr.ids; e.pts = fr.pts;
        st.events.push_back(std::move(e));
        st.n_frames++; frames_out++;
    }
}

struct GroupResult {
    int rc = 0;
    std::string err;
    long solved = 0;                // sequence-frames solved in steady state
    double seconds = 0;             // wall time of the steady-state part
    double step_ms_sum[6] = {0, 0, 0, 0, 0, 0};
    long steps = 0;
};

void run_group(const Stream &st, int first_seq, int n_seq, const std::string &out_dir, int write_upto, int feed_threads, bool resident, GroupResult &res) {
    isv_estimator_t *e = nullptr;
    res.rc = isv_estimator_create(&st.params, n_seq, &e);
    if (res.rc != ISV_OK) { res.err = "isv_estimator_create failed (a GPU is required)"; return; }
    // the windows stay on the device between frames (include/isvins_estimator.h); --no-resident: pack and upload them every frame
    if (resident && isv_estimator_set_resident(e, 1) != ISV_OK) { res.rc = ISV_ERR_UNSUPPORTED; res.err = isv_estimator_last_error(e); isv_estimator_destroy(e); return; }
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
    // every estimator owns two HIP streams; the runtime multiplexes a process's streams onto 4 hardware queues by default,
    // which serialises the groups' fork / join patterns against each other (256 sequences in 4 groups: 15.7 k frames/s
    // with 4 queues, 27.8 k with 16).  Ask for more before the runtime initialises, unless the caller has chosen.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    if (argc < 2) { fprintf(stderr, "usage: isv_replay STREAM | --euroc MAV0_DIR --tracks CSV --config TXT  [--sequences S] [--groups K] [--out DIR] [--write W] [--no-resident]\n"); return 2; }
    int S = 1, K = 0, W = 1;                        // K = 0: choose the groups from the sequence count
    bool resident = true;                           // windows kept on the device between frames (--no-resident: re-upload every frame)
    std::string out_dir, euroc_dir, tracks_path, config_path, stream_path, dump_path;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--dump-events") && i + 1 < argc) { dump_path = argv[++i]; continue; }
        if (!strcmp(argv[i], "--euroc") && i + 1 < argc) euroc_dir = argv[++i];
        else if (!strcmp(argv[i], "--tracks") && i + 1 < argc) tracks_path = argv[++i];
        else if (!strcmp(argv[i], "--config") && i + 1 < argc) config_path = argv[++i];
        else if (argv[i][0] != '-' && stream_path.empty()) stream_path = argv[i];
        else if (!strcmp(argv[i], "--sequences") && i + 1 < argc) S = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--groups") && i + 1 < argc) K = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_dir = argv[++i];
        else if (!strcmp(argv[i], "--write") && i + 1 < argc) W = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--no-resident")) resident = false;
        else if (!strcmp(argv[i], "--resident")) resident = true;
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (K == 0) { K = S / 512; if (K > 8) K = 8; if (K < 1) K = 1; }      // ~512 sequences per group, at most 8 groups
    if (S < 1 || K < 1 || K > S) { fprintf(stderr, "need 1 <= groups <= sequences\n"); return 2; }
    Stream st;
    std::string err;
    if (!euroc_dir.empty()) {
        if (tracks_path.empty() || config_path.empty()) { fprintf(stderr, "--euroc needs --tracks and --config\n"); return 2; }
        std::vector<ImuSample> imu;
        std::vector<Frame> frames;
        if (!read_stream(config_path.c_str(), st, err) || !read_imu_csv(euroc_dir + "/imu0/data.csv", imu, err) || !read_tracks_csv(tracks_path, frames, err)) { fprintf(stderr, "isv_replay: %s\n", err.c_str()); return 1; }
        Event boot_ev; const Event *boot = nullptr;      // CONFIG.txt may carry the `boot` rows behind its config line
        for (const Event &e : st.events) if (e.kind == 1) { boot_ev = e; boot = &boot_ev; }
        st.events.clear(); st.n_frames = 0;
        assemble_events(imu, frames, boot, st.params.cfg.n_frames, st);
    } else if (stream_path.empty() || !read_stream(stream_path.c_str(), st, err)) { fprintf(stderr, "isv_replay: %s\n", stream_path.empty() ? "no stream file" : err.c_str()); return 1; }
    if (!dump_path.empty()) {          // the processIMU / processImage calls this input turns into; no GPU needed
        FILE *f = fopen(dump_path.c_str(), "w");
        if (!f) { fprintf(stderr, "isv_replay: cannot write %s\n", dump_path.c_str()); return 1; }
        for (const Event &e : st.events) {
            if (e.kind == 0) fprintf(f, "imu %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", e.dt, e.acc[0], e.acc[1], e.acc[2], e.gyr[0], e.gyr[1], e.gyr[2]);
            else if (e.kind == 1) fprintf(f, "boot\n");
            else fprintf(f, "frame %.17g %zu\n", e.stamp, e.ids.size());
        }
        fclose(f);
        return 0;
    }
    int feed_threads = (int)std::thread::hardware_concurrency() / K;      // per group, for the per-sequence feed
    if (feed_threads > 8) feed_threads = 8;
    if (feed_threads < 1) feed_threads = 1;
    // the library's own per-sequence host work (isv_estimator_step, isv_batch_upload) uses min(8, cores) threads PER CALL: with K
    // groups calling at once that oversubscribes the host (4 groups x 8 threads on 16 cores); give every group its share
    { char buf[16]; snprintf(buf, sizeof(buf), "%d", feed_threads); setenv("ISV_HOST_THREADS", buf, 0); }
    std::vector<GroupResult> res(K);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < K; k++) {
        const int a = (int)((long)S * k / K), b = (int)((long)S * (k + 1) / K);
        th.emplace_back(run_group, std::cref(st), a, b - a, out_dir, W, feed_threads, resident, std::ref(res[k]));
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
    printf("{\"resident_windows\": %s, \"sequences\": %d, \"groups\": %d, \"frames_in_stream\": %d, \"n_frames_window\": %d, \"solved_in_steady_state\": %ld, "
           "\"steady_state_seconds\": %.6f, \"frames_per_second\": %.1f, \"wall_seconds\": %.3f, "
           "\"mean_step_ms\": {\"step\": %.3f, \"features_and_packing\": %.3f, \"triangulate\": %.3f, \"init_factor_graph\": %.3f, \"solve_odometry\": %.3f, \"readback_and_slide\": %.3f}}\n",
           resident ? "true" : "false", S, K, st.n_frames, st.params.cfg.n_frames, solved, secs, secs > 0 ? solved / secs : 0.0, wall,
           steps ? ms[0] / steps : 0.0, steps ? ms[1] / steps : 0.0, steps ? ms[2] / steps : 0.0, steps ? ms[3] / steps : 0.0, steps ? ms[4] / steps : 0.0,
           steps ? ms[5] / steps : 0.0);
    return 0;
}
