// isv_backend_impl.h -- the handle behind isv_backend_t, shared by the host translation units of the C ABI
// (isv_backend.hip: batches of caller windows; isv_sequence.hip: device-resident sequences).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "isv_device_types.h"
#include "isv_kernels.h"

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return ISV_ERR_DEVICE;                                                               \
        }                                                                                        \
    } while (0)

struct isv_backend {
    isv_config_t cfg;
    std::string err;
    hipStream_t stream = nullptr, stream2 = nullptr;
    hipEvent_t fj[4] = {};
    hipEvent_t pk[2] = {};        // isv_batch_pack_results: handle stream -> caller stream -> handle stream
    hipEvent_t ev[8] = {};
    std::vector<hipEvent_t> prof_ev;      // [max_iter][ISV_PROF_FAMILIES][2]
    int prof_valid = 0;
    DevBatch d{};                 // device pointers
    SolverHost hc;                // device figures + environment hooks, read once at creation
    std::vector<void *> allocs;
    // capacities
    size_t capB = 0, capL = 0, capF = 0, capTiles = 0;
    // host staging (pinned)
    struct Host {
        double *Ps, *Rs, *Vs, *Bas, *Bgs, *tic, *ric, *depth, *lm_pts_i, *f_pts_j, *f_pts_z, *imu_in, *imu_cov;
        int32_t *lm_off, *f_off, *lm_host, *lm_k, *lm_f0, *tile_win, *tile_f0, *tile_n, *imu_skip, *n_rp, *solve_flag, *pg_perm, *pg_off, *pg_sched, *pg_sched_off;
        FactorRec *f_rec;
        int32_t *pg_rec, *pg_wstart; double *pg_pts;
        int32_t *lm_optr; double *obs_raw;       // raw CSR of the upload (round 5: the derived arrays are built on the device)
        uint32_t *lm_meta; int32_t *margin_old; double *header0;
        isv_se3_prior_t *se3; isv_linear9_t *lin9; isv_relpose_t *relpose; isv_rollpitch_t *rollpitch;
        SolveState *st;
        double *pose, *sb, *ex, *lam;
    } h{};
    SolverStage stage{};          // pinned staging of the result records
    std::vector<void *> hallocs;
    // pristine copies for isv_batch_optimize restore
    double *Ps0 = nullptr, *Rs0 = nullptr, *Vs0 = nullptr, *Bas0 = nullptr, *Bgs0 = nullptr, *depth0 = nullptr, *tic0 = nullptr, *ric0 = nullptr;
    isv_se3_prior_t *se30 = nullptr; isv_linear9_t *lin90 = nullptr; isv_relpose_t *relpose0 = nullptr; isv_rollpitch_t *rollpitch0 = nullptr;
    void *arena_h = nullptr, *arena_d = nullptr; size_t arena_bytes = 0;      // the raw upload's arrays, one pinned and one device block of the same layout (one copy per upload)
    // (round 5) isv_batch_download in TWO copies: the window's state / priors / depths sit next to each other inside the upload block
    // (down_a_off .. + down_a_bytes), the solver's outputs (tangent state, depth flags, traces, marginalisation records) in a block of their own
    size_t down_a_off = 0, down_a_bytes = 0;
    void *down_h = nullptr, *down_d = nullptr; size_t down_bytes = 0;
    int32_t *d_optr = nullptr; double *d_obs_raw = nullptr;     // device copies of the raw CSR (isv_batch_upload -> k_upload_build)
    bool dev_build = false;       // this handle derives the solver's view of an upload on the device (ISV_HOST_PACK=1: the host packer, for A/B and the bitwise test)
    int resident = 0;
    int device = 0;               // the HIP device the handle was created on; every entry point re-selects it
    double *init_scratch = nullptr, *init_kld = nullptr;   // initFactorGraph scratch, allocated on first use and kept
    size_t init_cap = 0;
    double last_ms[8] = {};
    int64_t last_counts[8] = {};
    hipGraphExec_t graph_exec = nullptr;    // ISV_GRAPH=1 (measurement hook): the captured launch chain of isv_batch_optimize
    uint64_t graph_key = 0;
    int64_t graph_counts[8] = {};
    void *seq = nullptr;          // device-resident sequences (isv_sequence.hip), allocated by isv_backend_seq_enable
    void (*seq_free)(void *) = nullptr;
};

template <typename T>
static inline int dalloc(isv_backend *h, T **p, size_t n) {
    void *q = nullptr;
    HIPCHK(h, hipMalloc(&q, (n ? n : 1) * sizeof(T)));
    h->allocs.push_back(q);
    *p = (T *)q;
    return ISV_OK;
}
template <typename T>
static inline int halloc(isv_backend *h, T **p, size_t n) {
    void *q = nullptr;
    HIPCHK(h, hipHostMalloc(&q, (n ? n : 1) * sizeof(T), hipHostMallocDefault));
    h->hallocs.push_back(q);
    *p = (T *)q;
    return ISV_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_ != ISV_OK) return rc_; } while (0)
// a handle's buffers and streams live on the device it was created on; a caller may drive it from any thread
// (fresh threads start on device 0), so every entry point selects that device first
#define ENTER(h) HIPCHK(h, hipSetDevice((h)->device))

