// isv_estimator.cpp -- the reference's host-side window manager for S sequences in lock step (include/isvins_estimator.h).
// Host C++ only: every solve goes through the three backend entry points in isv_solver_vtbl_t, which
// isv_estimator_create binds to the HIP backend (isv_backend_triangulate / _init_factor_graph / _optimize_batch).
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <cstdlib>
#include <string>
#include <thread>
#include <cstdint>
#include <vector>
#include "../../include/isvins_estimator.h"

namespace {

using V3 = std::array<double, 3>;
using M3 = std::array<double, 9>;      // row-major

inline V3 add(const V3 &a, const V3 &b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline V3 sub(const V3 &a, const V3 &b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline V3 mul(const V3 &a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }
inline V3 mv(const M3 &A, const V3 &x) {
    return {A[0] * x[0] + A[1] * x[1] + A[2] * x[2], A[3] * x[0] + A[4] * x[1] + A[5] * x[2], A[6] * x[0] + A[7] * x[1] + A[8] * x[2]};
}
inline V3 mtv(const M3 &A, const V3 &x) {      // A^T x
    return {A[0] * x[0] + A[3] * x[1] + A[6] * x[2], A[1] * x[0] + A[4] * x[1] + A[7] * x[2], A[2] * x[0] + A[5] * x[1] + A[8] * x[2]};
}
inline M3 mm(const M3 &A, const M3 &B) {
    M3 C;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
    return C;
}
inline M3 hat(const V3 &v) { return {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}; }
inline V3 cross(const V3 &a, const V3 &b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }

struct Quat { double w, x, y, z; };
inline Quat qmul(const Quat &a, const Quat &b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
// Eigen's q * v for a quaternion that need not be unit: v + 2w (u x v) + 2 u x (u x v)
inline V3 qrot(const Quat &q, const V3 &v) {
    const V3 u = {q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = add(uv, uv);
    return add(add(v, mul(uv, q.w)), cross(u, uv));
}
// Eigen's toRotationMatrix(), applied as is to a quaternion that need not be unit
inline M3 qmat(const Quat &q) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    return {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
}
// Eigen::Quaterniond(Matrix3d)
inline Quat quat_of(const M3 &R) {
    const double t = R[0] + R[4] + R[8];
    Quat q;
    if (t > 0) {
        double s = std::sqrt(t + 1.0);
        q.w = 0.5 * s; s = 0.5 / s;
        q.x = (R[7] - R[5]) * s; q.y = (R[2] - R[6]) * s; q.z = (R[3] - R[1]) * s;
        return q;
    }
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    double v[3];
    v[i] = 0.5 * s; s = 0.5 / s;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * s;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
    return q;
}

struct Sample { double dt; V3 acc, gyr; };

// IntegrationBase: the POD the factors read (isv_imu_t) + the sample the next step starts from
struct PreIntegration {
    isv_imu_t pod;
    V3 acc_0, gyr_0;
    double n2[4];        // ACC_N^2, GYR_N^2, ACC_W^2, GYR_W^2
    PreIntegration(const V3 &a0, const V3 &g0, const V3 &ba, const V3 &bg, const isv_estimator_params_t &p) : acc_0(a0), gyr_0(g0) {
        std::memset(&pod, 0, sizeof(pod));
        pod.delta_q[3] = 1.0;
        for (int i = 0; i < 15; i++) pod.jacobian[i * 15 + i] = 1.0;
        for (int k = 0; k < 3; k++) { pod.linearized_ba[k] = ba[k]; pod.linearized_bg[k] = bg[k]; }
        n2[0] = p.acc_n * p.acc_n; n2[1] = p.gyr_n * p.gyr_n; n2[2] = p.acc_w * p.acc_w; n2[3] = p.gyr_w * p.gyr_w;
    }
    // push_back -> propagate -> midPointIntegration (integration_base.h:31-158)
    // (AVX2 clone picked at load time where the CPU has it; no FMA contraction, so both clones round identically)
    __attribute__((target_clones("avx2", "default"))) void push_back(double dt, const V3 &acc_1, const V3 &gyr_1) {
        const V3 ba = {pod.linearized_ba[0], pod.linearized_ba[1], pod.linearized_ba[2]};
        const V3 bg = {pod.linearized_bg[0], pod.linearized_bg[1], pod.linearized_bg[2]};
        const Quat dq = {pod.delta_q[3], pod.delta_q[0], pod.delta_q[1], pod.delta_q[2]};
        const V3 a0 = sub(acc_0, ba), a1 = sub(acc_1, ba);
        const V3 w = sub(mul(add(gyr_0, gyr_1), 0.5), bg);
        const V3 un_acc_0 = qrot(dq, a0);
        const Quat rq = qmul(dq, Quat{1, w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2});
        const V3 un_acc_1 = qrot(rq, a1);
        const V3 un_acc = mul(add(un_acc_0, un_acc_1), 0.5);
        V3 rp, rv;
        for (int k = 0; k < 3; k++) {
            rp[k] = pod.delta_p[k] + pod.delta_v[k] * dt + 0.5 * un_acc[k] * dt * dt;
            rv[k] = pod.delta_v[k] + un_acc[k] * dt;
        }
        // F (15x15) and V (15x18) in 3x3 blocks
        const M3 Rd = qmat(dq), Rr = qmat(rq), Wx = hat(w);
        M3 ImW;
        for (int k = 0; k < 9; k++) ImW[k] = -Wx[k] * dt;
        ImW[0] += 1; ImW[4] += 1; ImW[8] += 1;
        const M3 RdA0 = mm(Rd, hat(a0)), RrA1 = mm(Rr, hat(a1)), RrA1W = mm(RrA1, ImW);
        double F[225] = {0.0}, V[15 * 18] = {0.0};
        auto f = [&](int r, int c, int a, int b) -> double & { return F[(r + a) * 15 + c + b]; };
        auto v = [&](int r, int c, int a, int b) -> double & { return V[(r + a) * 18 + c + b]; };
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            const int k = a * 3 + b;
            const double I = a == b ? 1.0 : 0.0;
            f(0, 0, a, b) = I;
            f(0, 3, a, b) = -0.25 * RdA0[k] * dt * dt + -0.25 * RrA1W[k] * dt * dt;
            f(0, 6, a, b) = I * dt;
            f(0, 9, a, b) = -0.25 * (Rd[k] + Rr[k]) * dt * dt;
            f(0, 12, a, b) = -0.25 * RrA1[k] * dt * dt * -dt;
            f(3, 3, a, b) = ImW[k];
            f(3, 12, a, b) = -1.0 * I * dt;
            f(6, 3, a, b) = -0.5 * RdA0[k] * dt + -0.5 * RrA1W[k] * dt;
            f(6, 6, a, b) = I;
            f(6, 9, a, b) = -0.5 * (Rd[k] + Rr[k]) * dt;
            f(6, 12, a, b) = -0.5 * RrA1[k] * dt * -dt;
            f(9, 9, a, b) = I;
            f(12, 12, a, b) = I;
            v(0, 0, a, b) = 0.25 * Rd[k] * dt * dt;
            v(0, 3, a, b) = 0.25 * -RrA1[k] * dt * dt * 0.5 * dt;
            v(0, 6, a, b) = 0.25 * Rr[k] * dt * dt;
            v(0, 9, a, b) = v(0, 3, a, b);
            v(3, 3, a, b) = 0.5 * I * dt;
            v(3, 9, a, b) = 0.5 * I * dt;
            v(6, 0, a, b) = 0.5 * Rd[k] * dt;
            v(6, 3, a, b) = 0.5 * -RrA1[k] * dt * 0.5 * dt;
            v(6, 6, a, b) = 0.5 * Rr[k] * dt;
            v(6, 9, a, b) = v(6, 3, a, b);
            v(9, 12, a, b) = I * dt;
            v(12, 15, a, b) = I * dt;
        }
        // jacobian = F jacobian;  covariance = F covariance F^T + V noise V^T.  F and V are block-sparse: only the 3x3
        // blocks listed per block row are non-zero (the others are exact zeros, which add nothing to a dense sum), so
        // the sums run over those columns only, in ascending order like the dense products
        static const int FN[5] = {5, 2, 4, 1, 1}, FC0[5][5] = {{0, 3, 6, 9, 12}, {3, 12, 0, 0, 0}, {3, 6, 9, 12, 0}, {9, 0, 0, 0, 0}, {12, 0, 0, 0, 0}};
        static const int VN[5] = {4, 2, 4, 1, 1}, VC0[5][4] = {{0, 3, 6, 9}, {3, 9, 0, 0}, {0, 3, 6, 9}, {12, 0, 0, 0}, {15, 0, 0, 0}};
        // (row-times-matrix "axpy" form: the inner loops run over a contiguous row and vectorise; every entry still
        // accumulates its terms in ascending k)
        double Jn[225] = {0.0}, FC[225] = {0.0}, Cn[225] = {0.0}, Ft[225], Vt[18 * 15];
        for (int i = 0; i < 15; i++) for (int k = 0; k < 15; k++) Ft[k * 15 + i] = F[i * 15 + k];
        for (int i = 0; i < 15; i++) for (int k = 0; k < 18; k++) Vt[k * 15 + i] = V[i * 18 + k];
        const double nd[6] = {n2[0], n2[1], n2[0], n2[1], n2[2], n2[3]};
        for (int i = 0; i < 15; i++) {
            const int rb = i / 3;
            double *jn = &Jn[i * 15], *fc = &FC[i * 15], *cn = &Cn[i * 15];
            for (int b = 0; b < FN[rb]; b++)
                for (int k = FC0[rb][b]; k < FC0[rb][b] + 3; k++) {
                    const double f = F[i * 15 + k];
                    const double *jr = &pod.jacobian[k * 15], *cr = &pod.covariance[k * 15];
                    for (int j = 0; j < 15; j++) { jn[j] += f * jr[j]; fc[j] += f * cr[j]; }
                }
            for (int k = 0; k < 15; k++) {
                const double f = fc[k];
                const double *fr = &Ft[k * 15];
                for (int j = 0; j < 15; j++) cn[j] += f * fr[j];
            }
            double q[15] = {0.0};
            for (int b = 0; b < VN[rb]; b++)
                for (int k = VC0[rb][b]; k < VC0[rb][b] + 3; k++) {
                    const double vn = V[i * 18 + k] * nd[k / 3];
                    const double *vr = &Vt[k * 15];
                    for (int j = 0; j < 15; j++) q[j] += vn * vr[j];
                }
            for (int j = 0; j < 15; j++) cn[j] += q[j];
        }
        std::memcpy(pod.jacobian, Jn, sizeof(Jn)); std::memcpy(pod.covariance, Cn, sizeof(Cn));
        const double nq = std::sqrt(rq.w * rq.w + rq.x * rq.x + rq.y * rq.y + rq.z * rq.z);      // delta_q.normalize()
        pod.delta_q[0] = rq.x / nq; pod.delta_q[1] = rq.y / nq; pod.delta_q[2] = rq.z / nq; pod.delta_q[3] = rq.w / nq;
        for (int k = 0; k < 3; k++) { pod.delta_p[k] = rp[k]; pod.delta_v[k] = rv[k]; }
        pod.sum_dt += dt;
        acc_0 = acc_1; gyr_0 = gyr_1;
    }
};

// IDFeatures (include/feature_tracker/feature_manager.h:44-63).  The per-frame points (Feature::point of frames
// start_frame, start_frame + 1, ...) live in the sequence's pool: slot `slot` is a ring of POINT_RING entries that
// starts at `off` -- dropping the oldest observation is off++, and a track record stays 32 bytes, cheap to compact.
constexpr int POINT_RING = 32;             // >= the longest window (ISV_MAX_FRAMES)
struct Track {
    int id, start_frame;
    int slot, off, n;              // ring slot in Sequence::pool, first element, number of observations
    int solve_flag = 0;
    double depth = -1.0;           // estimated_depth
    int end_frame() const { return start_frame + n - 1; }
};

enum Flag { INITIAL = 0, NON_LINEAR = 1, INITIAL_STRUCTURE = 2 };

struct Sequence {
    int N = 0, Nvo = 0;
    std::vector<V3> Ps, Vs, Bas, Bgs;
    std::vector<M3> Rs;
    std::vector<double> Headers;
    std::vector<std::unique_ptr<PreIntegration>> pre;
    std::vector<std::vector<Sample>> bufs;
    int frame_count = 0;
    bool first_imu = true;
    V3 acc_0{}, gyr_0{};
    Flag flag = INITIAL;
    std::vector<Track> tracks;                 // f_manager.IDsfeatures, insertion order
    std::vector<V3> pool;                      // POINT_RING points per slot
    std::vector<int> free_slots;
    std::vector<int> hkey, hval;               // scratch: open-addressing id -> track index of addFeatureAndCheckParallax
    const V3 &pt(const Track &t, int i) const { return pool[(size_t)t.slot * POINT_RING + ((t.off + i) & (POINT_RING - 1))]; }
    V3 &pt(const Track &t, int i) { return pool[(size_t)t.slot * POINT_RING + ((t.off + i) & (POINT_RING - 1))]; }
    int new_slot() {
        if (!free_slots.empty()) { const int k = free_slots.back(); free_slots.pop_back(); return k; }
        const int k = (int)(pool.size() / POINT_RING);
        pool.resize(pool.size() + POINT_RING);
        return k;
    }
    void push_point(Track &t, const V3 &v) { pt(t, t.n) = v; t.n++; }
    void pop_front(Track &t) { t.off = (t.off + 1) & (POINT_RING - 1); t.n--; }
    void erase_point(Track &t, int i) { for (int k = i; k + 1 < t.n; k++) pt(t, k) = pt(t, k + 1); t.n--; }
    // keep the tracks for which keep(t) is true, in order; the others give their slot back
    template <class Pred> void compact(Pred keep) {
        size_t o = 0;
        for (size_t i = 0; i < tracks.size(); i++) {
            if (keep(tracks[i])) { if (o != i) tracks[o] = tracks[i]; o++; }
            else free_slots.push_back(tracks[i].slot);
        }
        tracks.resize(o);
    }
    isv_se3_prior_t pose_prior{};
    isv_linear9_t vb_prior{};
    std::vector<isv_relpose_t> relpose;        // edge (i, i + 1) = vioRelativePoseEdges[i + 1]
    std::vector<isv_rollpitch_t> rollpitch;    // vioRollPitchEdges
    bool margin_old = true;
    bool have_to_add = false;
    isv_se3_prior_t add_pose_prior{};          // forwardPosePriorEdgeToAdd
    isv_relpose_t add_relpose{};               // backwardRelativePoseEdgeToAdd
    isv_linear9_t add_vb{};                    // backwardVBEdgeToAdd
    // staged input
    bool staged = false;
    bool features_added = false;      // the resident path already ran addFeatureAndCheckParallax for the staged image and then fell back to the host path (same frame)
    int tracks_before = 0;            // tracks before that call: what the device's track list holds
    double staged_header = 0;
    std::vector<std::pair<int, V3>> staged_image;
    bool have_boot = false;
    std::vector<V3> boot_P, boot_V;
    std::vector<M3> boot_R;
    // the ABI view handed to the backend
    isv_window_t w{};
    std::vector<double> wPs, wRs, wVs, wBas, wBgs, wobs, wdepth, wpose, wsb, wex, wfeat;
    double wtic[3], wric[9];
    double cur_tic[3] = {0, 0, 0}, cur_ric[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // tic[0] / ric[0] as the last solve left them (cfg.estimate_extrinsic = 1 only)
    bool have_ex = false;
    std::vector<int32_t> wstart, wptr, wflag;
    std::vector<isv_imu_t> wimu;
    std::vector<isv_relpose_t> wrel;
    std::vector<isv_rollpitch_t> wrp;
    isv_se3_prior_t wpp{};
    isv_linear9_t wvb{};
    std::vector<int> good;                     // indices into tracks of the landmarks in `w`
    // device-resident mode (isv_estimator_set_resident): the window lives in slot `index of the sequence` of the backend
    bool resident = false;
    int last_slide = 0;                        // what slide_window did after the previous solve: 1 MARGIN_OLD, 2 MARGIN_SECOND_NEW (0: already on the device)
    std::vector<isv_seq_obs_t> frame_obs;      // the newest frame's observations as (track ordinal, slot, point)
    std::vector<int32_t> frame_flags;          // lm_solve_flag of the good landmarks, from the device
    isv_imu_t frame_imu[2];
    // outputs
    isv_summary_t last_summary{};
    int n_solves = 0, n_good_last = 0, n_failed = 0;   // n_failed: solves whose result was not finite (state kept as it was)
    std::vector<std::array<double, 8>> pose_rows;
    std::vector<std::array<double, 13>> newest_rows;
};

}  // namespace

struct isv_estimator {
    isv_estimator_params_t p{};
    isv_solver_vtbl_t solver{};
    isv_backend_t *backend = nullptr;          // owned when created by isv_estimator_create
    std::vector<Sequence> seq;
    std::string err;
    double step_ms[6] = {0, 0, 0, 0, 0, 0};
    bool resident_mode = false;                // asked for by isv_estimator_set_resident
    bool resident_ready = false;               // every sequence is seeded on the device
    int tracks_cap = 0;
    int64_t resident_frames = 0;
    int64_t resident_fallbacks = 0;            // frames the resident path handed to the re-upload path (isv_estimator_resident_fallbacks)
    // test hooks, read ONCE at creation (getenv on every step of every estimator is neither cheap nor safe beside a setenv):
    // ISV_DEBUG_SEQ_UNSUPPORTED_FRAME / ISV_DEBUG_SEQ_PRECHECK_FAIL_FRAME = the resident frame that "does not fit" (after / before the
    // feature pass); each fires once
    int64_t dbg_unsupported_frame = -1, dbg_precheck_frame = -1;
    bool dbg_unsupported_fired = false, dbg_precheck_fired = false;
};

namespace {

// Utility::deltaQ(theta).toRotationMatrix(): q = [1, theta / 2], not normalised
M3 delta_q_matrix(const V3 &theta) { return qmat(Quat{1.0, theta[0] / 2, theta[1] / 2, theta[2] / 2}); }

// FeatureManager::addFeatureAndCheckParallax (feature_manager.cpp:52-101) + compensatedParallax2 (:356-390)
bool add_features(Sequence &s, double min_parallax, bool record = false) {
    const int fc = s.frame_count;
    if (record) s.frame_obs.clear();
    // id -> track index: a flat open-addressing table rebuilt per frame (the reference does a linear find_if per feature)
    size_t cap = 64;
    while (cap < 2 * (s.tracks.size() + s.staged_image.size()) + 8) cap <<= 1;
    s.hkey.assign(cap, 0); s.hval.assign(cap, -1);
    auto slot_of = [&](int id) { size_t h = ((uint32_t)id * 2654435761u) & (cap - 1); while (s.hval[h] >= 0 && s.hkey[h] != id) h = (h + 1) & (cap - 1); return h; };
    for (size_t i = 0; i < s.tracks.size(); i++) { const size_t h = slot_of(s.tracks[i].id); s.hkey[h] = s.tracks[i].id; s.hval[h] = (int)i; }
    std::sort(s.staged_image.begin(), s.staged_image.end(), [](const std::pair<int, V3> &a, const std::pair<int, V3> &b) { return a.first < b.first; });
    int last_track_num = 0;
    for (const auto &ob : s.staged_image) {
        const size_t h = slot_of(ob.first);
        if (s.hval[h] < 0) {
            Track t; t.id = ob.first; t.start_frame = fc; t.slot = s.new_slot(); t.off = 0; t.n = 0;
            s.push_point(t, ob.second);
            s.hkey[h] = ob.first; s.hval[h] = (int)s.tracks.size();
            if (record) s.frame_obs.push_back(isv_seq_obs_t{(int32_t)s.tracks.size(), t.slot, {ob.second[0], ob.second[1], ob.second[2]}});
            s.tracks.push_back(t);
        } else {
            Track &t = s.tracks[s.hval[h]];
            if (t.n < POINT_RING) {      // (a second observation of an id in one image would overflow a window-long track)
                s.push_point(t, ob.second);
                if (record) s.frame_obs.push_back(isv_seq_obs_t{(int32_t)s.hval[h], t.slot, {ob.second[0], ob.second[1], ob.second[2]}});
            }
            last_track_num++;
        }
    }
    if (fc < 2 || last_track_num < 20) return true;
    double parallax_sum = 0;
    int parallax_num = 0;
    for (const Track &t : s.tracks) {
        if (t.start_frame <= fc - 2 && t.end_frame() >= fc - 1) {
            const V3 &pi = s.pt(t, fc - 2 - t.start_frame), &pj = s.pt(t, fc - 1 - t.start_frame);
            const double du = pi[0] / pi[2] - pj[0], dv = pi[1] / pi[2] - pj[1];
            parallax_sum += std::max(0.0, std::sqrt(du * du + dv * dv));
            parallax_num++;
        }
    }
    if (parallax_num == 0) return true;
    return parallax_sum / parallax_num >= min_parallax;
}

// the Estimator members backendOptimization() touches, as an isv_window_t over this sequence's buffers
int build_window(const isv_estimator *e, Sequence &s, std::string &err) {
    const int N = s.N, Nvo = s.Nvo;
    s.good.clear();
    size_t n_obs = 0;
    for (size_t i = 0; i < s.tracks.size(); i++)            // goodFeature(): used_num >= 2 && start_frame < Vo_SIZE
        if (s.tracks[i].n >= 2 && s.tracks[i].start_frame < Nvo) { s.good.push_back((int)i); n_obs += (size_t)s.tracks[i].n; }
    const size_t L = s.good.size();
    if ((int)L > e->p.cfg.max_landmarks || (int)n_obs > e->p.cfg.max_obs) { err = "window exceeds the landmark / observation capacity"; return ISV_ERR_CAPACITY; }
    if ((int)s.rollpitch.size() > e->p.cfg.max_rollpitch) { err = "more roll/pitch factors than max_rollpitch"; return ISV_ERR_CAPACITY; }
    s.wPs.resize(N * 3); s.wRs.resize(N * 9); s.wVs.resize(N * 3); s.wBas.resize(N * 3); s.wBgs.resize(N * 3);
    for (int i = 0; i < N; i++) {
        std::memcpy(&s.wPs[i * 3], s.Ps[i].data(), 24); std::memcpy(&s.wRs[i * 9], s.Rs[i].data(), 72); std::memcpy(&s.wVs[i * 3], s.Vs[i].data(), 24);
        std::memcpy(&s.wBas[i * 3], s.Bas[i].data(), 24); std::memcpy(&s.wBgs[i * 3], s.Bgs[i].data(), 24);
    }
    // estimate_extrinsic = 0: the configured extrinsic, every frame (a constant parameter block; the reference's double2vector
    // round-trips it through a quaternion each frame, a rounding-level drift this port does not reproduce).  = 1: what the
    // previous solve left in tic[0] / ric[0] (double2vector, src/estimator.cpp:575-583), the configured one before the first solve.
    if (e->p.cfg.estimate_extrinsic && s.have_ex) { std::memcpy(s.wtic, s.cur_tic, 24); std::memcpy(s.wric, s.cur_ric, 72); }
    else { std::memcpy(s.wtic, e->p.tic, 24); std::memcpy(s.wric, e->p.ric, 72); }
    s.wstart.resize(std::max<size_t>(L, 1)); s.wptr.resize(L + 1); s.wflag.assign(std::max<size_t>(L, 1), 0);
    s.wobs.resize(std::max<size_t>(n_obs, 1) * 3); s.wdepth.resize(std::max<size_t>(L, 1)); s.wfeat.resize(std::max<size_t>(L, 1));
    s.wptr[0] = 0;
    size_t o = 0;
    for (size_t l = 0; l < L; l++) {
        const Track &t = s.tracks[s.good[l]];
        s.wstart[l] = t.start_frame; s.wdepth[l] = t.depth;
        for (int k = 0; k < t.n; k++) { std::memcpy(&s.wobs[o * 3], s.pt(t, k).data(), 24); o++; }
        s.wptr[l + 1] = (int32_t)o;
    }
    s.wimu.resize(N - 1);
    for (int j = 1; j < N; j++) {
        if (!s.pre[j]) { err = "a window frame has no pre-integration (no IMU samples were fed)"; return ISV_ERR_INVALID_ARG; }
        s.wimu[j - 1] = s.pre[j]->pod;
    }
    s.wpp = s.pose_prior; s.wvb = s.vb_prior;
    s.wrel = s.relpose;
    s.wrp = s.rollpitch;
    if (s.wrp.empty()) s.wrp.resize(1);
    s.wpose.assign(N * 7, 0.0); s.wsb.assign(N * 9, 0.0); s.wex.assign(7, 0.0);
    isv_window_t &w = s.w;
    w.Ps = s.wPs.data(); w.Rs = s.wRs.data(); w.Vs = s.wVs.data(); w.Bas = s.wBas.data(); w.Bgs = s.wBgs.data();
    w.tic = s.wtic; w.ric = s.wric;
    w.n_landmarks = (int32_t)L; w.n_obs = (int32_t)n_obs;
    w.lm_start_frame = s.wstart.data(); w.lm_obs_ptr = s.wptr.data(); w.obs_point = s.wobs.data();
    w.lm_depth = s.wdepth.data(); w.lm_solve_flag = s.wflag.data();
    w.imu = s.wimu.data();
    w.pose_prior = &s.wpp; w.vb_prior = &s.wvb; w.relpose = s.wrel.data(); w.rollpitch = s.wrp.data();
    w.n_rollpitch = (int32_t)s.rollpitch.size();
    w.margin_old = s.margin_old ? 1 : 0;
    w.header0 = s.Headers[0];
    w.para_Pose = s.wpose.data(); w.para_SpeedBias = s.wsb.data(); w.para_Ex_Pose = s.wex.data(); w.para_Feature = s.wfeat.data();
    return ISV_OK;
}

// what double2vector() and update() leave in the Estimator members
void read_back(Sequence &s) {
    const int N = s.N;
    for (int i = 0; i < N; i++) {
        std::memcpy(s.Ps[i].data(), &s.wPs[i * 3], 24); std::memcpy(s.Rs[i].data(), &s.wRs[i * 9], 72); std::memcpy(s.Vs[i].data(), &s.wVs[i * 3], 24);
        std::memcpy(s.Bas[i].data(), &s.wBas[i * 3], 24); std::memcpy(s.Bgs[i].data(), &s.wBgs[i * 3], 24);
    }
    for (size_t l = 0; l < s.good.size(); l++) { Track &t = s.tracks[s.good[l]]; t.depth = s.wdepth[l]; t.solve_flag = s.wflag[l]; }
    s.pose_prior = s.wpp; s.vb_prior = s.wvb;
    s.relpose = s.wrel;
    for (size_t i = 0; i < s.rollpitch.size(); i++) s.rollpitch[i] = s.wrp[i];
    std::memcpy(s.cur_tic, s.wtic, 24); std::memcpy(s.cur_ric, s.wric, 72); s.have_ex = true;      // (double2vector: tic[0], ric[0])
}

void new_preintegration(const isv_estimator *e, Sequence &s, int j) {
    s.pre[j].reset(new PreIntegration(s.acc_0, s.gyr_0, s.Bas[j], s.Bgs[j], e->p));
}

// Estimator::slideWindow  src/estimator.cpp:1565-1724
void slide_window(const isv_estimator *e, Sequence &s) {
    const int N = s.N, Nvo = s.Nvo;
    if (s.margin_old) {
        const M3 back_R0 = s.Rs[0];
        const V3 back_P0 = s.Ps[0];
        if (s.frame_count != N - 1) return;
        for (int i = 0; i < N - 1; i++) {
            std::swap(s.Ps[i], s.Ps[i + 1]); std::swap(s.Rs[i], s.Rs[i + 1]); std::swap(s.Vs[i], s.Vs[i + 1]);
            std::swap(s.Bas[i], s.Bas[i + 1]); std::swap(s.Bgs[i], s.Bgs[i + 1]); std::swap(s.Headers[i], s.Headers[i + 1]);
            std::swap(s.pre[i], s.pre[i + 1]); std::swap(s.bufs[i], s.bufs[i + 1]);
        }
        s.Headers[N - 1] = s.Headers[N - 2];
        s.Ps[N - 1] = s.Ps[N - 2]; s.Rs[N - 1] = s.Rs[N - 2]; s.Vs[N - 1] = s.Vs[N - 2]; s.Bas[N - 1] = s.Bas[N - 2]; s.Bgs[N - 1] = s.Bgs[N - 2];
        new_preintegration(e, s, N - 1);
        s.bufs[N - 1].clear();
        const bool shift_depth = s.flag == NON_LINEAR;
        if (shift_depth && s.have_to_add) {            // the prior factors move one frame towards the past (:1607-1645)
            for (auto &f : s.relpose) { f.imu_i -= 1; f.imu_j -= 1; }                  // RelativePoseFactor::shift()
            s.relpose.erase(s.relpose.begin());
            s.add_relpose.imu_i = Nvo - 2; s.add_relpose.imu_j = Nvo - 1;
            s.relpose.push_back(s.add_relpose);
            std::vector<isv_rollpitch_t> kept;
            for (auto &f : s.rollpitch) { f.index -= 1; if (f.index >= 0) kept.push_back(f); }      // RollPitchFactor::shift()
            s.rollpitch.swap(kept);
            s.add_pose_prior.index = 0; s.pose_prior = s.add_pose_prior;
            s.add_vb.index = Nvo - 1; s.vb_prior = s.add_vb;
            s.have_to_add = false;
        }
        if (shift_depth) {                             // slideWindowOld -> removeBackShiftDepth  feature_manager.cpp:275-313
            // ric[0] / tic[0] of slideWindowOld (src/estimator.cpp:1714-1719): the configured extrinsic, or the estimated one
            const bool est = e->p.cfg.estimate_extrinsic && s.have_ex;
            const double *rr = est ? s.cur_ric : e->p.ric, *tt = est ? s.cur_tic : e->p.tic;
            const M3 ric = {rr[0], rr[1], rr[2], rr[3], rr[4], rr[5], rr[6], rr[7], rr[8]};
            const V3 tic = {tt[0], tt[1], tt[2]};
            const M3 R0 = mm(back_R0, ric), R1 = mm(s.Rs[0], ric);
            const V3 P0 = add(back_P0, mv(back_R0, tic)), P1 = add(s.Ps[0], mv(s.Rs[0], tic));
            s.compact([&](Track &t) {
                if (t.start_frame != 0) { t.start_frame--; return true; }
                const V3 uv = s.pt(t, 0);
                s.pop_front(t);
                if (t.n < 2) return false;
                if (s.resident) return true;        // (the depths and window states live on the device: k_seq_slide re-hosts them)
                const V3 w_pt = add(mv(R0, mul(uv, t.depth)), P0);
                const V3 pj = mtv(R1, sub(w_pt, P1));
                t.depth = pj[2] > 0 ? pj[2] : e->p.cfg.init_depth;
                return true;
            });
        } else {                                       // removeBack  :315-332
            s.compact([&](Track &t) {
                if (t.start_frame != 0) { t.start_frame--; return true; }
                s.pop_front(t);
                return t.n > 0;
            });
        }
    } else {
        const int fc = s.frame_count;
        if (fc != N - 1) return;
        for (const Sample &smp : s.bufs[fc]) {
            s.pre[fc - 1]->push_back(smp.dt, smp.acc, smp.gyr);
            s.bufs[fc - 1].push_back(smp);
        }
        s.Headers[fc - 1] = s.Headers[fc];
        s.Ps[fc - 1] = s.Ps[fc]; s.Rs[fc - 1] = s.Rs[fc]; s.Vs[fc - 1] = s.Vs[fc]; s.Bas[fc - 1] = s.Bas[fc]; s.Bgs[fc - 1] = s.Bgs[fc];
        new_preintegration(e, s, N - 1);
        s.bufs[N - 1].clear();
        s.compact([&](Track &t) {                      // slideWindowNew -> removeFront  :335-354
            if (t.start_frame == fc) { t.start_frame--; return true; }
            if (t.end_frame() < fc - 1) return true;
            s.erase_point(t, N - 1 - 1 - t.start_frame);
            return t.n > 0;
        });
    }
}

void after_solve(const isv_estimator *e, Sequence &s, double header) {
    slide_window(e, s);
    s.compact([](Track &t) { return t.solve_flag != 2; });      // removeFailures
    const int N = s.N;
    std::array<double, 13> nr;
    nr[0] = header;
    std::memcpy(&nr[1], s.Ps[N - 1].data(), 24); std::memcpy(&nr[4], s.Rs[N - 1].data(), 72);
    s.newest_rows.push_back(nr);
    const Quat q = quat_of(s.Rs[0]);
    s.pose_rows.push_back({s.Headers[0], s.Ps[0][0], s.Ps[0][1], s.Ps[0][2], q.w, q.x, q.y, q.z});
}

// run body(i) for i in [0, n) on min(8, cores) host threads (ISV_HOST_THREADS overrides), one per >= 8 items; the
// sequences are independent of each other.  Returns the first non-zero status.
template <class Body>
int parallel_for(int n, std::vector<std::string> &errs, Body body) {
    int K = 0;
    if (const char *ev = getenv("ISV_HOST_THREADS")) K = atoi(ev);
    if (K <= 0) { K = (int)std::thread::hardware_concurrency(); if (K > 8) K = 8; if (K > n / 8) K = n / 8; }
    if (K > n) K = n;
    if (K < 1) K = 1;
    std::vector<int> rcs(K, ISV_OK);
    errs.assign(K, std::string());
    auto work = [&](int k) {
        for (int i = (int)((int64_t)n * k / K), end = (int)((int64_t)n * (k + 1) / K); i < end; i++) {
            const int rc = body(i, errs[k]);
            if (rc != ISV_OK) { rcs[k] = rc; return; }
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < K; k++) th.emplace_back(work, k);
    work(0);
    for (auto &t : th) t.join();
    for (int k = 0; k < K; k++) if (rcs[k] != ISV_OK) { if (k) errs[0] = errs[k]; return rcs[k]; }
    return ISV_OK;
}

int hip_triangulate(void *ctx, int32_t n, isv_window_t *const *w) { return isv_backend_triangulate((isv_backend_t *)ctx, n, w); }
int hip_init(void *ctx, isv_window_t *w, isv_summary_t *s, double *kld) { return isv_backend_init_factor_graph((isv_backend_t *)ctx, w, s, kld); }
int hip_optimize(void *ctx, int32_t n, isv_window_t *const *w, isv_summary_t *s, isv_marg_result_t *m) {
    return isv_backend_optimize_batch((isv_backend_t *)ctx, n, w, s, m);
}

int hip_init_batch(void *ctx, int32_t n, isv_window_t *const *w, isv_summary_t *s, double *kld) {
    return isv_backend_init_factor_graph_batch((isv_backend_t *)ctx, n, w, s, kld);
}
int hip_solve_odometry(void *ctx, int32_t n, isv_window_t *const *w, isv_summary_t *s, isv_marg_result_t *m) {
    return isv_backend_solve_odometry_batch((isv_backend_t *)ctx, n, w, s, m);
}

int create_common(const isv_estimator_params_t *p, int32_t n_sequences, isv_estimator **out) {
    if (!p || !out || n_sequences < 1) return ISV_ERR_INVALID_ARG;
    const int N = p->cfg.n_frames, Nvo = p->cfg.n_vo;
    if (N < 3 || N > POINT_RING || Nvo < 2 || Nvo > N - 1) return ISV_ERR_INVALID_ARG;
    isv_estimator *e = new isv_estimator();
    e->p = *p;
    e->p.cfg.max_batch = n_sequences;
    if (const char *ev = getenv("ISV_DEBUG_SEQ_UNSUPPORTED_FRAME")) e->dbg_unsupported_frame = strtoll(ev, nullptr, 10);
    if (const char *ev = getenv("ISV_DEBUG_SEQ_PRECHECK_FAIL_FRAME")) e->dbg_precheck_frame = strtoll(ev, nullptr, 10);
    e->seq.resize(n_sequences);
    for (Sequence &s : e->seq) {
        s.N = N; s.Nvo = Nvo;
        s.Ps.assign(N, V3{0, 0, 0}); s.Vs = s.Ps; s.Bas = s.Ps; s.Bgs = s.Ps;
        s.Rs.assign(N, M3{1, 0, 0, 0, 1, 0, 0, 0, 1});
        s.Headers.assign(N, 0.0);
        s.pre.resize(N); s.bufs.resize(N);
        s.relpose.assign(Nvo - 1, isv_relpose_t{});
    }
    *out = e;
    return ISV_OK;
}

}  // namespace

extern "C" int isv_estimator_create(const isv_estimator_params_t *p, int32_t n_sequences, isv_estimator_t **out) {
    if (out) *out = nullptr;
    isv_estimator *e = nullptr;
    int rc = create_common(p, n_sequences, &e);
    if (rc != ISV_OK) return rc;
    rc = isv_backend_create(&e->p.cfg, &e->backend);           // fails loudly without a GPU: there is no other solver
    if (rc != ISV_OK) { delete e; return rc; }
    e->solver = isv_solver_vtbl_t{e->backend, hip_triangulate, hip_init, hip_optimize, hip_init_batch, hip_solve_odometry};
    *out = e;
    return ISV_OK;
}

extern "C" int isv_estimator_create_with_solver(const isv_estimator_params_t *p, int32_t n_sequences, const isv_solver_vtbl_t *solver,
                                                isv_estimator_t **out) {
    if (out) *out = nullptr;
    if (!solver || !solver->triangulate || !solver->init_factor_graph || !solver->optimize_batch) return ISV_ERR_INVALID_ARG;
    isv_estimator *e = nullptr;
    const int rc = create_common(p, n_sequences, &e);
    if (rc != ISV_OK) return rc;
    e->solver = *solver;
    *out = e;
    return ISV_OK;
}

extern "C" void isv_estimator_destroy(isv_estimator_t *e) {
    if (!e) return;
    if (e->backend) isv_backend_destroy(e->backend);
    delete e;
}

extern "C" const char *isv_estimator_last_error(const isv_estimator_t *e) { return e ? e->err.c_str() : "null handle"; }

#define SEQ_OR_FAIL(e, seq) \
    if (!(e) || (seq) < 0 || (size_t)(seq) >= (e)->seq.size()) return ISV_ERR_INVALID_ARG

// Estimator::processIMU  src/estimator.cpp:91-124
extern "C" int isv_estimator_process_imu(isv_estimator_t *e, int32_t seq, double dt, const double acc[3], const double gyr[3]) {
    SEQ_OR_FAIL(e, seq);
    if (!acc || !gyr) return ISV_ERR_INVALID_ARG;
    Sequence &s = e->seq[seq];
    const V3 a = {acc[0], acc[1], acc[2]}, g = {gyr[0], gyr[1], gyr[2]};
    if (s.first_imu) { s.first_imu = false; s.acc_0 = a; s.gyr_0 = g; }
    const int j = s.frame_count;
    if (!s.pre[j]) new_preintegration(e, s, j);
    if (j != 0) {
        s.pre[j]->push_back(dt, a, g);
        s.bufs[j].push_back(Sample{dt, a, g});
        const V3 G = {e->p.cfg.gravity[0], e->p.cfg.gravity[1], e->p.cfg.gravity[2]};
        const V3 un_acc_0 = sub(mv(s.Rs[j], sub(s.acc_0, s.Bas[j])), G);
        const V3 un_gyr = sub(mul(add(s.gyr_0, g), 0.5), s.Bgs[j]);
        s.Rs[j] = mm(s.Rs[j], delta_q_matrix(mul(un_gyr, dt)));
        const V3 un_acc_1 = sub(mv(s.Rs[j], sub(a, s.Bas[j])), G);
        const V3 un_acc = mul(add(un_acc_0, un_acc_1), 0.5);
        for (int k = 0; k < 3; k++) {
            s.Ps[j][k] = s.Ps[j][k] + dt * s.Vs[j][k] + 0.5 * dt * dt * un_acc[k];
            s.Vs[j][k] = s.Vs[j][k] + dt * un_acc[k];
        }
    }
    s.acc_0 = a; s.gyr_0 = g;
    return ISV_OK;
}

extern "C" int isv_estimator_process_imu_n(isv_estimator_t *e, int32_t seq, int32_t n, const double *dt, const double *acc, const double *gyr) {
    SEQ_OR_FAIL(e, seq);
    if (n < 0 || (n > 0 && (!dt || !acc || !gyr))) return ISV_ERR_INVALID_ARG;
    for (int i = 0; i < n; i++) {
        const int rc = isv_estimator_process_imu(e, seq, dt[i], acc + i * 3, gyr + i * 3);
        if (rc != ISV_OK) return rc;
    }
    return ISV_OK;
}

extern "C" int isv_estimator_last_step_ms(const isv_estimator_t *e, double out[6]) {
    if (!e || !out) return ISV_ERR_INVALID_ARG;
    std::memcpy(out, e->step_ms, sizeof(e->step_ms));
    return ISV_OK;
}

extern "C" int isv_estimator_push_image(isv_estimator_t *e, int32_t seq, double header, int32_t n, const int32_t *feature_id,
                                        const double *point) {
    SEQ_OR_FAIL(e, seq);
    if (n < 0 || (n > 0 && (!feature_id || !point))) return ISV_ERR_INVALID_ARG;
    Sequence &s = e->seq[seq];
    if (s.staged) { e->err = "an image is already staged for this sequence"; return ISV_ERR_INVALID_ARG; }
    s.staged_image.clear();
    s.staged_image.reserve(n);
    for (int i = 0; i < n; i++) s.staged_image.emplace_back(feature_id[i], V3{point[i * 3], point[i * 3 + 1], point[i * 3 + 2]});
    // a feature id seen twice in one image: the reference's std::map<int, vector<...>> keeps both and processImage reads the FIRST
    // (id_pts.second[0], src/feature_tracker/feature_manager.cpp:64-66); the resident track store appends one observation per
    // track and frame (k_seq_append), so duplicates are dropped here, once, for both paths
    std::stable_sort(s.staged_image.begin(), s.staged_image.end(), [](const std::pair<int, V3> &a, const std::pair<int, V3> &b) { return a.first < b.first; });
    s.staged_image.erase(std::unique(s.staged_image.begin(), s.staged_image.end(), [](const std::pair<int, V3> &a, const std::pair<int, V3> &b) { return a.first == b.first; }), s.staged_image.end());
    s.staged_header = header;
    s.staged = true;
    return ISV_OK;
}

extern "C" int isv_estimator_set_bootstrap(isv_estimator_t *e, int32_t seq, const double *Ps, const double *Rs, const double *Vs) {
    SEQ_OR_FAIL(e, seq);
    if (!Ps || !Rs || !Vs) return ISV_ERR_INVALID_ARG;
    Sequence &s = e->seq[seq];
    s.boot_P.resize(s.N); s.boot_V.resize(s.N); s.boot_R.resize(s.N);
    for (int i = 0; i < s.N; i++) {
        std::memcpy(s.boot_P[i].data(), Ps + i * 3, 24); std::memcpy(s.boot_V[i].data(), Vs + i * 3, 24); std::memcpy(s.boot_R[i].data(), Rs + i * 9, 72);
    }
    s.have_boot = true;
    return ISV_OK;
}

namespace {

// every sequence's window -> the backend's resident slots (after a slide, i.e. between two frames)
int seed_resident(isv_estimator *e) {
    const int S = (int)e->seq.size();
    std::vector<isv_window_t *> ws(S);
    std::vector<int32_t> nt(S);
    std::vector<std::vector<isv_seq_track_t>> trk(S);
    std::vector<std::vector<double>> pts(S);
    std::vector<const isv_seq_track_t *> trk_p(S);
    std::vector<const double *> pts_p(S);
    for (int si = 0; si < S; si++) {
        Sequence &s = e->seq[si];
        if ((int)s.tracks.size() > e->tracks_cap) return ISV_ERR_CAPACITY;
        std::string err;
        const int rc = build_window(e, s, err);
        if (rc != ISV_OK) { e->err = err; return rc; }
        ws[si] = &s.w;
        nt[si] = (int32_t)s.tracks.size();
        trk[si].resize(s.tracks.size());
        for (size_t i = 0; i < s.tracks.size(); i++) {
            const Track &t = s.tracks[i];
            if (t.slot >= e->tracks_cap) return ISV_ERR_CAPACITY;
            trk[si][i] = isv_seq_track_t{t.start_frame, t.n, t.solve_flag, t.slot, t.depth};
            for (int k = 0; k < t.n; k++) { const V3 &p = s.pt(t, k); pts[si].insert(pts[si].end(), p.begin(), p.end()); }
        }
        if (pts[si].empty()) pts[si].resize(3);
        trk_p[si] = trk[si].data(); pts_p[si] = pts[si].data();
    }
    const int rc = isv_backend_seq_seed(e->backend, S, ws.data(), nt.data(), trk_p.data(), pts_p.data());
    if (rc != ISV_OK) { e->err = std::string("seeding the resident windows failed: ") + isv_backend_last_error(e->backend); return rc; }
    for (Sequence &s : e->seq) { s.resident = true; s.last_slide = 0; }
    e->resident_ready = true;
    return ISV_OK;
}

// the device's state of every sequence back into the host members (a sequence leaves the resident mode BEFORE its slide)
// (pre_add: the host has ALREADY appended this frame's features to its track lists -- the device holds the first tracks_before
//  tracks of every sequence; the tracks behind them are this frame's new ones and keep their fresh depth)
// (slid: per sequence, overrides host_has_slid -- a frame in which only SOME sequences had an image: the idle ones' slide of their
//  last solve is still pending on the device while the active ones have not slid yet)
int leave_resident(isv_estimator *e, bool host_has_slid, bool pre_add = false, const std::vector<char> *slid = nullptr) {
    auto ntracks = [&](const Sequence &s) { return pre_add ? (size_t)s.tracks_before : s.tracks.size(); };
    auto has_slid = [&](size_t si) { return slid ? (*slid)[si] != 0 : host_has_slid; };
    bool any_slid = false;
    for (size_t si = 0; si < e->seq.size(); si++) any_slid |= has_slid(si);
    if (any_slid) {           // the device applies a slide with the NEXT frame: let it catch up with the host's bookkeeping first
        std::vector<int32_t> prev(e->seq.size()), nt(e->seq.size());
        for (size_t si = 0; si < e->seq.size(); si++) { prev[si] = has_slid(si) ? e->seq[si].last_slide : 0; nt[si] = (int32_t)ntracks(e->seq[si]); }
        const int rc = isv_backend_seq_flush(e->backend, (int32_t)e->seq.size(), prev.data(), nt.data());
        if (rc != ISV_OK) { e->err = std::string("leaving the resident mode failed: ") + isv_backend_last_error(e->backend); return rc; }
        for (size_t si = 0; si < e->seq.size(); si++) if (has_slid(si)) e->seq[si].last_slide = 0;
    }
    for (size_t si = 0; si < e->seq.size(); si++) {
        Sequence &s = e->seq[si];
        if (!s.resident) continue;
        const int N = s.N;
        s.wPs.resize(N * 3); s.wRs.resize(N * 9); s.wVs.resize(N * 3); s.wBas.resize(N * 3); s.wBgs.resize(N * 3);
        s.wrel.resize(s.Nvo - 1); s.wrp.resize(std::max(1, e->p.cfg.max_rollpitch));
        isv_window_t w{};
        w.Ps = s.wPs.data(); w.Rs = s.wRs.data(); w.Vs = s.wVs.data(); w.Bas = s.wBas.data(); w.Bgs = s.wBgs.data();
        w.pose_prior = &s.wpp; w.vb_prior = &s.wvb; w.relpose = s.wrel.data(); w.rollpitch = s.wrp.data();
        w.tic = s.wtic; w.ric = s.wric;
        const size_t ntr = ntracks(s);
        std::vector<double> dep(std::max<size_t>(ntr, 1));
        std::vector<int32_t> fl(std::max<size_t>(ntr, 1));
        const int rc = isv_backend_seq_download(e->backend, (int32_t)si, &w, (int32_t)ntr, dep.data(), fl.data());
        if (rc != ISV_OK) { e->err = std::string("leaving the resident mode failed: ") + isv_backend_last_error(e->backend); return rc; }
        // (after a slide the newest frame is the host's: processIMU may already have propagated it with the next frame's samples)
        for (int i = 0; i < (has_slid(si) ? N - 1 : N); i++) {
            std::memcpy(s.Ps[i].data(), &s.wPs[i * 3], 24); std::memcpy(s.Rs[i].data(), &s.wRs[i * 9], 72); std::memcpy(s.Vs[i].data(), &s.wVs[i * 3], 24);
            std::memcpy(s.Bas[i].data(), &s.wBas[i * 3], 24); std::memcpy(s.Bgs[i].data(), &s.wBgs[i * 3], 24);
        }
        if (e->p.cfg.estimate_extrinsic) { std::memcpy(s.cur_tic, s.wtic, 24); std::memcpy(s.cur_ric, s.wric, 72); s.have_ex = true; }     // (tic[0] / ric[0])
        s.pose_prior = s.wpp; s.vb_prior = s.wvb; s.relpose = s.wrel;
        s.rollpitch.assign(s.wrp.begin(), s.wrp.begin() + w.n_rollpitch);
        for (size_t i = 0; i < ntr; i++) { s.tracks[i].depth = dep[i]; s.tracks[i].solve_flag = fl[i]; }
        s.resident = false;
    }
    e->resident_ready = false;
    return ISV_OK;
}

#define RESIDENT_FELL_BACK (-1000)      // resident_frame: the frame did not fit the resident path; the windows are back on the host
// one lock-step frame of the resident sequences: only what is new crosses PCIe (include/isvins_backend.h)
int resident_frame(isv_estimator *e, std::vector<std::string> &errs) {
    const int S = (int)e->seq.size();
    const auto tr0 = std::chrono::steady_clock::now();
    std::vector<isv_seq_frame_t> fr(S);
    std::vector<isv_seq_result_t> res(S);
    std::vector<int32_t *> flags(S);
    std::vector<char> fits(S, 1), idle(S, 0);
    //  what can REFUSE the frame is checked for every sequence before any of them is touched: an error return must not
    // leave some sequences with this image's features in their track lists and the device without them
    for (const Sequence &s : e->seq)
        if (s.staged && (!s.pre[s.N - 1] || (s.last_slide == 2 && !s.pre[s.N - 2]))) { e->err = "a window frame has no pre-integration (no IMU samples were fed)"; return ISV_ERR_INVALID_ARG; }
    int rc = parallel_for(S, errs, [&](int si, std::string &err) {
        Sequence &s = e->seq[si];
        const int N = s.N;
        isv_seq_frame_t &f = fr[si];
        std::memset(&f, 0, sizeof(f));
        if (!s.staged) {                           // no image for this sequence this step (round 4): it idles on the device, its slide stays pending
            f.prev_slide = -1; f.n_tracks = (int32_t)s.tracks.size();
            s.tracks_before = (int)s.tracks.size();
            idle[si] = 1; flags[si] = nullptr;
            return (int)ISV_OK;
        }
        f.prev_slide = s.last_slide;
        f.n_tracks = (int32_t)s.tracks.size();
        // (from here on the host state changes.  A window that does not fit the resident store or the per-window
        //  kernels is NOT an error -- the re-upload path solves it -- so the limits only mark the frame for the fall-back below;
        //  the tracks this call appends are remembered so that the device's shorter list can be brought back consistently)
        s.tracks_before = (int)s.tracks.size();
        s.margin_old = add_features(s, e->p.min_parallax, true);
        s.Headers[s.frame_count] = s.staged_header;
        s.staged = false; s.features_added = true;
        if ((int)s.tracks.size() > e->tracks_cap) fits[si] = 0;
        f.margin_old = s.margin_old ? 1 : 0;
        f.n_obs = (int32_t)s.frame_obs.size(); f.obs = s.frame_obs.data();
        for (const isv_seq_obs_t &o : s.frame_obs) if (o.slot >= e->tracks_cap) fits[si] = 0;
        if (s.last_slide == 2) { s.frame_imu[0] = s.pre[N - 2]->pod; s.frame_imu[1] = s.pre[N - 1]->pod; f.n_imu = 2; }
        else { s.frame_imu[0] = s.pre[N - 1]->pod; f.n_imu = 1; }
        f.imu = s.frame_imu;
        std::memcpy(f.Ps, s.Ps[N - 1].data(), 24); std::memcpy(f.Rs, s.Rs[N - 1].data(), 72); std::memcpy(f.Vs, s.Vs[N - 1].data(), 24);
        std::memcpy(f.Bas, s.Bas[N - 1].data(), 24); std::memcpy(f.Bgs, s.Bgs[N - 1].data(), 24);
        f.header0 = s.Headers[0];
        s.good.clear();
        int64_t n_obs = 0;
        for (size_t i = 0; i < s.tracks.size(); i++)
            if (s.tracks[i].n >= 2 && s.tracks[i].start_frame < s.Nvo) { s.good.push_back((int)i); n_obs += s.tracks[i].n; }
        f.n_landmarks = (int32_t)s.good.size(); f.n_factors = (int32_t)(n_obs - (int64_t)s.good.size());
        if (f.n_landmarks > e->p.cfg.max_landmarks || n_obs > e->p.cfg.max_obs) fits[si] = 0;      // (the host path reports it, as it always did)
        s.frame_flags.assign(std::max<size_t>(s.good.size(), 1), 0);
        flags[si] = s.frame_flags.data();
        return (int)ISV_OK;
    });
    if (rc != ISV_OK) { e->err = errs[0]; return rc; }
    const auto tr1 = std::chrono::steady_clock::now();
    bool fit_all = true;
    for (int si = 0; si < S; si++) fit_all &= fits[si] != 0;
    if (!e->dbg_unsupported_fired && e->resident_frames == e->dbg_unsupported_frame) { fit_all = false; e->dbg_unsupported_fired = true; }      // (test hook, once)
    // isv_backend_seq_frame refuses a frame BEFORE it launches anything (capacity, or windows the per-window kernels do not take):
    // the device still holds the state of the previous solve
    rc = fit_all ? isv_backend_seq_frame(e->backend, S, fr.data(), res.data(), flags.data(), nullptr) : (int)ISV_ERR_UNSUPPORTED;
    if (rc == ISV_ERR_CAPACITY || rc == ISV_ERR_UNSUPPORTED) {
        const int rcl = leave_resident(e, true, true);
        if (rcl != ISV_OK) return rcl;
        e->resident_fallbacks++;
        return RESIDENT_FELL_BACK;                   // isv_estimator_step goes on with the host path for this frame (the features are in)
    }
    if (rc != ISV_OK) { e->err = std::string("resident frame failed: ") + isv_backend_last_error(e->backend); return rc; }
    for (Sequence &s : e->seq) s.features_added = false;
    int n_active = 0;
    for (int si = 0; si < S; si++) n_active += !idle[si];
    const auto tr2 = std::chrono::steady_clock::now();
    e->resident_frames++;
    bool failed = false;
    for (int si = 0; si < S; si++) failed |= !idle[si] && res[si].summary.status != ISV_OK;
    if (failed) {
        // a non-finite solve: everything comes back to the host (the device has not slid the sequences of this frame yet; an idle one's
        // slide of its last solve is still pending and is applied first) and the host path's policy applies
        rc = leave_resident(e, false, false, &idle);
        if (rc != ISV_OK) return rc;
        for (int si = 0; si < S; si++) {          // the sequences that did solve still install their marginalisation outputs
            Sequence &s = e->seq[si];
            isv_marg_result_t m;
            if (idle[si] || res[si].summary.status != ISV_OK || !s.margin_old) continue;
            rc = isv_backend_seq_marg(e->backend, si, &m);
            if (rc != ISV_OK) { e->err = isv_backend_last_error(e->backend); return rc; }
            if (!m.valid) continue;
            s.add_pose_prior = m.forward_pose_prior; s.add_relpose = m.backward_relpose; s.add_vb = m.backward_vb; s.have_to_add = true;
            isv_rollpitch_t brp = m.backward_rollpitch; brp.index = s.Nvo - 1;
            s.rollpitch.push_back(brp);
        }
    }
    (void)parallel_for(S, errs, [&](int si, std::string &) {
        Sequence &s = e->seq[si];
        if (idle[si]) return (int)ISV_OK;          // (nothing was solved for it: no slide, no row, its counters stand)
        const int N = s.N;
        const isv_seq_result_t &r = res[si];
        const bool ok = r.summary.status == ISV_OK;
        if (!failed) {      // the frames the host reads: newest (processIMU), oldest and second (pose_output.txt rows)
            std::memcpy(s.Ps[N - 1].data(), r.Ps_new, 24); std::memcpy(s.Rs[N - 1].data(), r.Rs_new, 72); std::memcpy(s.Vs[N - 1].data(), r.Vs_new, 24);
            std::memcpy(s.Bas[N - 1].data(), r.Bas_new, 24); std::memcpy(s.Bgs[N - 1].data(), r.Bgs_new, 24);
            std::memcpy(s.Ps[0].data(), r.Ps_old, 24); std::memcpy(s.Rs[0].data(), r.Rs_old, 72);
            std::memcpy(s.Ps[1].data(), r.Ps_second, 24); std::memcpy(s.Rs[1].data(), r.Rs_second, 72);
            for (size_t l = 0; l < s.good.size(); l++) s.tracks[s.good[l]].solve_flag = s.frame_flags[l];
            if (e->p.cfg.estimate_extrinsic) { std::memcpy(s.cur_tic, r.tic, 24); std::memcpy(s.cur_ric, r.ric, 72); s.have_ex = true; }
        } else if (!ok) { s.n_failed++; s.have_to_add = false; }
        s.last_summary = r.summary;
        s.n_solves++;
        s.n_good_last = (int)s.good.size();
        after_solve(e, s, s.Headers[N - 1]);
        s.last_slide = s.margin_old ? 1 : 2;
        if (failed && !ok) { s.flag = INITIAL_STRUCTURE; s.rollpitch.clear(); }
        return (int)ISV_OK;
    });
    auto msd = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    e->step_ms[1] = msd(tr0, tr1); e->step_ms[4] = msd(tr1, tr2); e->step_ms[5] = msd(tr2, std::chrono::steady_clock::now());
    return n_active;
}

}  // namespace

// keep every sequence's window on the device between frames (include/isvins_backend.h, "device-resident sequences")
extern "C" int isv_estimator_set_resident(isv_estimator_t *e, int32_t on) {
    if (!e) return ISV_ERR_INVALID_ARG;
    if (!on) {
        if (e->resident_ready) { const int rc = leave_resident(e, true); if (rc != ISV_OK) return rc; }
        e->resident_mode = false;
        return ISV_OK;
    }
    if (!e->backend) { e->err = "the resident mode needs the HIP backend"; return ISV_ERR_UNSUPPORTED; }
    e->tracks_cap = std::max(64, 2 * e->p.cfg.max_landmarks);
    const int rc = isv_backend_seq_enable(e->backend, e->tracks_cap);
    if (rc != ISV_OK) { e->err = std::string("isv_backend_seq_enable: ") + isv_backend_last_error(e->backend); return rc; }
    e->resident_mode = true;
    return ISV_OK;
}
extern "C" int64_t isv_estimator_resident_frames(const isv_estimator_t *e) { return e ? e->resident_frames : 0; }
extern "C" int64_t isv_estimator_resident_fallbacks(const isv_estimator_t *e) { return e ? e->resident_fallbacks : -1; }

// Estimator::processImage (src/estimator.cpp:126-215) on every staged sequence, the solves batched
extern "C" int isv_estimator_step(isv_estimator_t *e) {
    if (!e) return ISV_ERR_INVALID_ARG;
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = clk::now();
    for (double &x : e->step_ms) x = 0;
    std::vector<std::string> errs;
    if (e->resident_ready) {
        // (round 4: a sequence WITHOUT an image this step no longer evicts the group -- it idles on the device, src/System.cpp:160-202
        //  pairs IMU and images per sequence -- but every sequence must be resident and in steady state)
        bool all = true, any = false;
        for (const Sequence &s : e->seq) { all &= s.resident && s.flag == NON_LINEAR; any |= s.staged; }
        if (!any) return 0;
        // the resident store's own limits, checked BEFORE anything is mutated.  First the cheap upper bound (every staged id taken as a
        // new track); a sequence that fails it is counted exactly -- only the ids it does not track yet open a track (a tight
        // handle that really fits used to fail the bound on every frame and fall back to the slower path without a word)
        bool fits = true;
        for (const Sequence &s : e->seq) {
            if (!s.staged) continue;
            auto ok = [&](size_t n_new) {
                const size_t fresh = n_new > s.free_slots.size() ? n_new - s.free_slots.size() : 0;
                return s.tracks.size() + n_new <= (size_t)e->tracks_cap && s.pool.size() / POINT_RING + fresh <= (size_t)e->tracks_cap;
            };
            if (ok(s.staged_image.size())) continue;
            std::vector<int> ids(s.tracks.size());
            for (size_t i = 0; i < s.tracks.size(); i++) ids[i] = s.tracks[i].id;
            std::sort(ids.begin(), ids.end());
            size_t n_new = 0;
            for (const auto &ob : s.staged_image) n_new += !std::binary_search(ids.begin(), ids.end(), ob.first);
            fits &= ok(n_new);
        }
        if (!e->dbg_precheck_fired && e->resident_frames == e->dbg_precheck_frame) { fits = false; e->dbg_precheck_fired = true; }      // (test hook, once)
        if (all && fits) {
            const int rc = resident_frame(e, errs);        // (fills step_ms[1] host preparation, [4] hand-over + device, [5] read-back + slide)
            e->step_ms[0] = ms(t0, clk::now());
            if (rc != RESIDENT_FELL_BACK) return rc;
            // the frame did not fit the resident path: the windows are back, this frame's features are already in the track lists
        } else {
            e->resident_fallbacks++;
            const int rc = leave_resident(e, true);    // a sequence left the steady state, or the store is full:
            if (rc != ISV_OK) return rc;               // the host path takes over (and seeds again once every sequence solves in one frame)
        }
    }
    // per sequence: addFeatureAndCheckParallax, Headers, the INITIAL bookkeeping; mark[si] = 1 when the sequence solves
    std::vector<char> mark(e->seq.size(), 0);
    int rc = parallel_for((int)e->seq.size(), errs, [&](int si, std::string &err) {
        Sequence &s = e->seq[si];
        if (!s.staged && !s.features_added) return (int)ISV_OK;
        if (s.flag == INITIAL && s.frame_count == s.N - 1 && !s.have_boot) { err = "the window is full: isv_estimator_set_bootstrap first"; return (int)ISV_ERR_INVALID_ARG; }
        if (!s.features_added) {                   // (else: the resident path ran addFeatureAndCheckParallax for this image before it fell back)
            s.margin_old = add_features(s, e->p.min_parallax);
            s.Headers[s.frame_count] = s.staged_header;
            s.staged = false;
        }
        s.features_added = false;
        if (s.flag == INITIAL) {
            if (s.frame_count == s.N - 1) {
                s.Ps = s.boot_P; s.Rs = s.boot_R; s.Vs = s.boot_V;       // in place of initialStructure()
                s.flag = INITIAL_STRUCTURE;
                mark[si] = 1;
            } else s.frame_count++;
        } else mark[si] = 1;
        return mark[si] ? build_window(e, s, err) : (int)ISV_OK;
    });
    if (rc != ISV_OK) { e->err = errs[0]; return rc; }
    std::vector<int> solve;
    for (size_t si = 0; si < e->seq.size(); si++) if (mark[si]) solve.push_back((int)si);
    if (solve.empty()) return 0;
    // solveOdometry (:461-472): f_manager.triangulate(Ps, tic, ric); backendOptimization()
    std::vector<isv_window_t *> ws(solve.size());
    for (size_t k = 0; k < solve.size(); k++) ws[k] = &e->seq[solve[k]].w;
    bool first_solve = false;
    for (int si : solve) first_solve |= e->seq[si].flag == INITIAL_STRUCTURE;
    std::vector<isv_summary_t> sums(solve.size());
    std::vector<isv_marg_result_t> margs(solve.size());
    const auto t1 = clk::now();
    auto t2 = t1, t3 = t1;
    if (!first_solve && e->solver.solve_odometry_batch) {
        rc = e->solver.solve_odometry_batch(e->solver.ctx, (int32_t)ws.size(), ws.data(), sums.data(), margs.data());
        if (rc != ISV_OK) { e->err = "solveOdometry failed"; return rc; }
    } else {
        rc = e->solver.triangulate(e->solver.ctx, (int32_t)ws.size(), ws.data());
        if (rc != ISV_OK) { e->err = "triangulate failed"; return rc; }
        for (int si : solve) {
            Sequence &s = e->seq[si];
            for (size_t l = 0; l < s.good.size(); l++) s.tracks[s.good[l]].depth = s.wdepth[l];
        }
        t2 = clk::now();
        // backendOptimization(), INITIAL_STRUCTURE branch (:1543-1548): vector2double, initFactorGraph, NON_LINEAR.  The
        // NON_LINEAR branch below runs in the same call (two `if`s in the reference, not else-if).
        std::vector<int> first;
        for (int si : solve) if (e->seq[si].flag == INITIAL_STRUCTURE) first.push_back(si);
        if (!first.empty() && e->solver.init_factor_graph_batch) {       // all first solves of this frame in one batch
            std::vector<isv_window_t *> wf(first.size());
            for (size_t k = 0; k < first.size(); k++) wf[k] = &e->seq[first[k]].w;
            std::vector<isv_summary_t> s0(first.size());
            rc = e->solver.init_factor_graph_batch(e->solver.ctx, (int32_t)wf.size(), wf.data(), s0.data(), nullptr);
            if (rc != ISV_OK) { e->err = "initFactorGraph failed"; return rc; }
        } else {
            for (int si : first) {
                isv_summary_t s0;
                double kld = 0;
                rc = e->solver.init_factor_graph(e->solver.ctx, &e->seq[si].w, &s0, &kld);
                if (rc != ISV_OK) { e->err = "initFactorGraph failed"; return rc; }
            }
        }
        for (int si : first) {
            Sequence &s = e->seq[si];
            s.rollpitch.clear();
            read_back(s);
            s.flag = NON_LINEAR;
            s.have_to_add = false;
            rc = build_window(e, s, e->err);
            if (rc != ISV_OK) return rc;
        }
        t3 = clk::now();
        rc = e->solver.optimize_batch(e->solver.ctx, (int32_t)ws.size(), ws.data(), sums.data(), margs.data());
        if (rc != ISV_OK) { e->err = "backendOptimization failed"; return rc; }
    }
    const auto t4 = clk::now();
    (void)parallel_for((int)solve.size(), errs, [&](int k, std::string &) {
        Sequence &s = e->seq[solve[k]];
        // A solve that produced a non-finite cost must not be copied into the window (the reference has no such guard:
        // its NaNs would spread through every later frame): the sequence keeps its pre-solve states and depths, drops
        // this frame's marginalisation outputs, and the failure is counted (isv_estimator_failed_solves).  The window
        // still slides (below), but without marginalisation outputs the prior factors cannot follow it -- with MARGIN_OLD
        // they would sit one frame off for every later solve -- so the sequence goes back to INITIAL_STRUCTURE after the
        // slide: its next solve rebuilds EVERY prior at the then-current states through initFactorGraph
        // (backendOptimization's first branch, src/estimator.cpp:1543-1548), exactly as after initialisation.
        const bool ok = sums[k].status == ISV_OK;
        if (ok) read_back(s); else { s.n_failed++; s.have_to_add = false; }
        const isv_marg_result_t &m = margs[k];
        if (ok && s.margin_old && m.valid) {
            s.add_pose_prior = m.forward_pose_prior; s.add_relpose = m.backward_relpose; s.add_vb = m.backward_vb;
            s.have_to_add = true;
            isv_rollpitch_t brp = m.backward_rollpitch;
            brp.index = s.Nvo - 1;
            s.rollpitch.push_back(brp);                 // vioRollPitchEdges.push_back  (MargBackward :1536-1538)
        }
        s.last_summary = sums[k];
        s.n_solves++;
        s.n_good_last = (int)s.good.size();
        after_solve(e, s, s.Headers[s.N - 1]);        // (slides as NON_LINEAR: removeBackShiftDepth re-hosts the depths)
        if (!ok) { s.flag = INITIAL_STRUCTURE; s.rollpitch.clear(); }
        return (int)ISV_OK;
    });
    if (e->resident_mode && !e->resident_ready && solve.size() == e->seq.size()) {
        // every sequence solved this frame and is in steady state: from the next frame on the windows stay on the device
        bool all = true;
        for (const Sequence &s : e->seq) all &= s.flag == NON_LINEAR && s.frame_count == s.N - 1 && !s.have_to_add;
        if (all) { const int rcs = seed_resident(e); if (rcs != ISV_OK && rcs != ISV_ERR_CAPACITY && rcs != ISV_ERR_UNSUPPORTED) return rcs; }
    }
    const auto t5 = clk::now();
    e->step_ms[0] = ms(t0, t5); e->step_ms[1] = ms(t0, t1); e->step_ms[2] = ms(t1, t2); e->step_ms[3] = ms(t2, t3);
    e->step_ms[4] = ms(t3, t4); e->step_ms[5] = ms(t4, t5);
    return (int)solve.size();
}

extern "C" int isv_estimator_status(const isv_estimator_t *e, int32_t seq, int32_t out[8]) {
    SEQ_OR_FAIL(e, seq);
    if (!out) return ISV_ERR_INVALID_ARG;
    const Sequence &s = e->seq[seq];
    out[0] = s.flag == NON_LINEAR ? 1 : 0; out[1] = s.frame_count; out[2] = s.margin_old ? 1 : 0; out[3] = (int32_t)s.tracks.size();
    out[4] = s.n_good_last; out[5] = (int32_t)s.rollpitch.size(); out[6] = s.n_solves; out[7] = s.last_summary.iterations;
    return ISV_OK;
}

extern "C" int isv_estimator_failed_solves(const isv_estimator_t *e, int32_t seq) {
    SEQ_OR_FAIL(e, seq);
    return e->seq[seq].n_failed;
}

extern "C" int isv_estimator_get_window(const isv_estimator_t *e, int32_t seq, double *Ps, double *Rs, double *Vs, double *Bas,
                                        double *Bgs, double *Headers) {
    SEQ_OR_FAIL(e, seq);
    const Sequence &s = e->seq[seq];
    if (s.resident) {
        // the window lives on the device and has NOT been slid yet there (the slide is applied with the next frame): only the
        // frames the host tracks itself (0, 1, N-1) are current in the members below
        ((isv_estimator *)e)->err = "isv_estimator_get_window: the sequence is device-resident; isv_estimator_set_resident(e, 0) first";
        return ISV_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < s.N; i++) {
        if (Ps) std::memcpy(Ps + i * 3, s.Ps[i].data(), 24);
        if (Rs) std::memcpy(Rs + i * 9, s.Rs[i].data(), 72);
        if (Vs) std::memcpy(Vs + i * 3, s.Vs[i].data(), 24);
        if (Bas) std::memcpy(Bas + i * 3, s.Bas[i].data(), 24);
        if (Bgs) std::memcpy(Bgs + i * 3, s.Bgs[i].data(), 24);
        if (Headers) Headers[i] = s.Headers[i];
    }
    return ISV_OK;
}

// tic[0] / ric[0] (the configured extrinsic, or with cfg.estimate_extrinsic = 1 what the last solve's double2vector left; current in
// the resident mode too: every resident frame's result record carries it)
extern "C" int isv_estimator_get_extrinsic(const isv_estimator_t *e, int32_t seq, double *tic, double *ric) {
    SEQ_OR_FAIL(e, seq);
    const Sequence &s = e->seq[seq];
    const bool est = e->p.cfg.estimate_extrinsic && s.have_ex;
    if (tic) std::memcpy(tic, est ? s.cur_tic : e->p.tic, 24);
    if (ric) std::memcpy(ric, est ? s.cur_ric : e->p.ric, 72);
    return ISV_OK;
}

extern "C" int isv_estimator_get_preintegration(const isv_estimator_t *e, int32_t seq, int32_t frame, isv_imu_t *out) {
    SEQ_OR_FAIL(e, seq);
    const Sequence &s = e->seq[seq];
    if (!out || frame < 0 || frame >= s.N || !s.pre[frame]) return ISV_ERR_INVALID_ARG;
    *out = s.pre[frame]->pod;
    return ISV_OK;
}

extern "C" int isv_estimator_last_summary(const isv_estimator_t *e, int32_t seq, isv_summary_t *out) {
    SEQ_OR_FAIL(e, seq);
    if (!out) return ISV_ERR_INVALID_ARG;
    *out = e->seq[seq].last_summary;
    return ISV_OK;
}

extern "C" int isv_estimator_trajectory(const isv_estimator_t *e, int32_t seq, int32_t which, double *out, int32_t max_rows) {
    SEQ_OR_FAIL(e, seq);
    const Sequence &s = e->seq[seq];
    if (which == 0) {
        for (int i = 0; out && i < max_rows && i < (int)s.pose_rows.size(); i++) std::memcpy(out + i * 8, s.pose_rows[i].data(), 64);
        return (int)s.pose_rows.size();
    }
    if (which == 1) {
        for (int i = 0; out && i < max_rows && i < (int)s.newest_rows.size(); i++) std::memcpy(out + i * 13, s.newest_rows[i].data(), 104);
        return (int)s.newest_rows.size();
    }
    return ISV_ERR_INVALID_ARG;
}
