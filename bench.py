#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X sliding-window VIO backend.

Metric (BASELINE.json): sliding-window solves/sec on synthetic 11-KF / ~300-landmark windows, plus
ms per optimize().  One STEP = one pass of the hot path (Estimator::backendOptimization():
vector2double + problemSolve [<= 10 dogleg iterations] + update() + double2vector + MargForward /
MargBackward) over the rank's device-resident batch of windows.  Inputs are resident in HBM before the
timed region (the step restores the pristine uploaded state on device first).

  python bench.py --gpus N --steps K --warmup W
With N > 1 and no RANK in the environment the script starts the N ranks itself (a torch.distributed.run
CHILD process, before anything in this process touches the GPU); under torch.distributed.run it is one
rank of the job.  Windows are independent, so ranks shard them with no collective inside the solve;
the ONE exchange step of the multi-sequence configuration -- an RCCL all-gather of the
per-window result records, device buffers, no host copy -- runs inside every timed step when N > 1.
  --scaling weak   (default) --windows per GPU (1024: BASELINE config 4's batch on every GPU)
  --scaling strong --windows in total, block-partitioned over the ranks (config 4 as written: 1024 / 8 = 128 per GPU)
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import isvins_loader  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector = matrix peak


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a C port of the reference algorithm), built on THIS host with -O3 -march=native into a
# temporary directory (never in-tree: a native build must not travel to another machine), 1 thread and all cores
def _native_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    src = os.path.join(ROOT, "oracle", "isv_oracle.c")
    so = os.path.join(tempfile.gettempdir(), f"libisv_oracle_native_{os.getuid()}_{os.getpid()}.so")
    flags = ["-O3", "-march=native", "-fPIC", "-std=gnu11", "-shared", "-pthread"]
    try:
        subprocess.check_call(["gcc", *flags, "-o", so, src, os.path.join(ROOT, "oracle", "isv_pgo_oracle.c"), "-lm"], stderr=subprocess.DEVNULL)
        lib = C.CDLL(so)
        os.unlink(so)
        how = "gcc -O3 -march=native"
    except Exception:
        lib = C.CDLL(os.path.join(ROOT, "oracle", "libisv_oracle.so")); how = "gcc -O2 (native build failed)"
    from isvins_amd import abi
    dp = C.POINTER(C.c_double)
    lib.isvo_optimize.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t), C.POINTER(abi.isv_summary_t), C.POINTER(abi.isv_marg_result_t)]
    lib.isvo_optimize.restype = C.c_int
    lib.isvo_linearize.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t), dp, dp, dp, dp]
    lib.isvo_linearize.restype = C.c_int
    lib.isvo_optimize_batch_mt.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(C.POINTER(abi.isv_window_t)), C.c_int, C.c_int,
                                           C.POINTER(abi.isv_summary_t), C.POINTER(abi.isv_marg_result_t)]
    lib.isvo_optimize_batch_mt.restype = C.c_int
    return lib, how


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(windows, cfg, budget_s=10.0):
    """full backendOptimization() per window on the host: 1 thread, then threads over windows on every core"""
    from isvins_amd import abi
    lib, how = _native_oracle()
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                   # a container's CPU quota (cgroup v2) bounds the cores this process really gets
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            nproc = max(1, min(nproc, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass

    def solve(w):
        s = abi.isv_summary_t(); mg = abi.isv_marg_result_t()
        lib.isvo_optimize(C.byref(cfg), C.byref(w.c()), C.byref(s), C.byref(mg))
        return s.iterations

    # 1 thread
    n1 = 0; t0 = time.perf_counter()
    for w in windows:
        solve(w.clone()); n1 += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt1 = time.perf_counter() - t0
    # all cores: pthreads over windows inside the C library (static block partition), marshalling done before the clock starts
    est = max(nproc, int(n1 / dt1 * budget_s * nproc * 0.7))
    sample = [windows[i % len(windows)].clone() for i in range(min(est, 8 * len(windows)))]
    n = len(sample)
    cs = [w.c() for w in sample]
    arr = (C.POINTER(abi.isv_window_t) * n)(*[C.pointer(c) for c in cs])
    sums = (abi.isv_summary_t * n)(); margs = (abi.isv_marg_result_t * n)()
    t0 = time.perf_counter()
    lib.isvo_optimize_batch_mt(C.byref(cfg), arr, n, nproc, sums, margs)
    dtn = time.perf_counter() - t0
    return {"value": n1 / dt1, "unit": "windows/s", "cores": 1, "kind": "port",
            "sample": f"{n1} of the benchmark's own 11-KF/300-landmark windows, full backendOptimization() each, oracle/isv_oracle.c built on this host ({how}), 1 thread",
            "ms_per_optimize": 1e3 * dt1 / n1,
            "all_cores": {"value": n / dtn, "unit": "windows/s", "cores": nproc,
                          "sample": f"{n} windows, {nproc} pthreads over windows (isvo_optimize_batch_mt) (the reference's problemSolve is single-threaded, src/estimator.cpp:1122: more cores only help across windows)"},
            "nproc": nproc, "cpu_model": _cpu_model()}, lib


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def bs_flops(N):
    """algorithmic flops of the structure-aware reduced-system solve of one window"""
    M = N // 2
    lo = lambda i: i - 1 if i > M else 0
    hi = lambda i: i + 1 if i < M else N - 1
    par = lambda i: i + 1 if i < M else (i - 1 if i > M else -1)
    fl = 0.0
    for i in range(N):
        nr = hi(i) - lo(i) + 1
        fl += 2 * 9 ** 3 / 6.0 * 2                                   # Cholesky + inverse of the 9x9 block
        fl += ((9 if par(i) >= 0 else 0) + 6 * nr + 1) * 45 * 2      # [C; Y; y^T] L^-T
        if par(i) >= 0:
            fl += (45 + 54 * nr + 9) * 9 * 2                         # parent downdates
    for I in range(N):
        for J in range(I + 1):
            cover = sum(1 for i in range(N) if lo(i) <= J and I <= hi(i))
            fl += cover * (21 if I == J else 36) * 9 * 2             # Spp -= Y Y^T
    for J in range(N):
        m = N - J - 1
        fl += 2 * 6 ** 3 / 6.0 * 2 + (6 * m + 1) * 21 * 2 + (m * (m + 1) / 2 * 36 + 6 * m) * 6 * 2
    fl += 2 * (6 * N) ** 2 + 4 * N * 81 * 2 + 4 * sum((hi(i) - lo(i) + 1) * 54 for i in range(N)) * 2   # triangular solves + gathers
    fl += 6.0 * (N * (N + 1) / 2 * 36 + 162 * N + sum((hi(i) - lo(i) + 1) * 54 for i in range(N)))     # scaling, u^T T u, LM diagonal
    return fl


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default: ~2 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=1024, help="windows per GPU (weak) or in total (strong); BASELINE config 4 batch")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--frames", type=int, default=11)
    ap.add_argument("--vo", type=int, default=5)
    ap.add_argument("--landmarks", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the untimed secondary legs (host-buffer rates, single window, config 2, N = 18, B = 128); used for the rocprofv3 passes so that their kernel statistics hold the benchmark's own launches only")
    args = ap.parse_args()

    # ---- N > 1 without a launcher: start the ranks as a CHILD job.  This process has not imported torch or touched the
    # GPU and never will (a process that initialised the GPU must not exec / spawn rank processes on this pool).
    if args.gpus > 1 and "RANK" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.scaling == "strong" and args.windows % world != 0:
        # equal shards only: all_gather_into_tensor needs the same record count on every rank
        sys.exit(f"bench.py: --scaling strong needs --windows ({args.windows}) divisible by the number of ranks ({world})")

    # more than two handles' worth of HIP streams in one process (the untimed two-handle leg below): the runtime maps
    # streams onto 4 hardware queues by default; must be set before the runtime initialises.  One handle is unaffected.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch
    # ISV_BENCH_REHEARSAL=1: rehearse the multi-rank path on ONE GPU (every rank on cuda:0, gloo instead of RCCL, which
    # refuses two ranks on one device); the driver's real runs have one GPU per rank
    rehearsal = os.environ.get("ISV_BENCH_REHEARSAL") == "1"
    dev = 0 if (world == 1 or rehearsal) else local_rank
    torch.cuda.set_device(dev)
    isvins_loader.load()
    from isvins_amd import backend, sharding, synth
    backend.build()

    W = args.windows if args.scaling == "weak" else None
    ids = list(sharding.shard_window_ids(rank, world, args.windows, None if args.scaling == "weak" else args.windows))
    W = len(ids)
    windows = synth.make_windows(ids, n_frames=args.frames, n_vo=args.vo, n_landmarks=args.landmarks)
    Ftot = sum(w.n_factors for w in windows)
    max_obs = max(w.n_obs for w in windows)
    cfg0 = backend.abi.make_config(args.frames, args.vo, max_landmarks=args.landmarks, max_obs=max_obs, max_batch=W)
    # CPU baseline on rank 0 BEFORE the process group exists: the other ranks sleep in the rendezvous meanwhile
    cpu = None; cpu_lib = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu, cpu_lib = cpu_baseline(windows, cfg0, budget_s=10.0 if world == 1 else 5.0)

    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if rehearsal else "nccl", timeout=datetime.timedelta(minutes=20))

    # The C-ABI library allocates on the calling thread's current HIP device (and every later entry point re-selects the
    # device the handle was created on).  backend.load_library() maps torch's HIP runtime first, so the library and torch
    # share ONE runtime; the free-memory check below verifies that the batch really landed on this rank's GPU.
    free_before = torch.cuda.mem_get_info(dev)[0]
    be = backend.Backend(args.frames, args.vo, max_landmarks=args.landmarks, max_obs=max_obs, max_batch=W)
    be.upload(windows)
    free_after = torch.cuda.mem_get_info(dev)[0]
    if W >= 64 and free_before - free_after < (64 << 20):
        raise RuntimeError(f"rank {rank}: the backend did not allocate on cuda:{dev} (free memory moved by {free_before - free_after} B)")

    rec = be.record_doubles()
    records = gathered = None
    if world > 1:
        gdev = "cpu" if rehearsal else f"cuda:{dev}"
        records = torch.zeros((W, rec), dtype=torch.float64, device=f"cuda:{dev}")
        gathered = torch.zeros((world * W, rec), dtype=torch.float64, device=gdev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        be.run_optimize(sync=False)
        if world > 1:      # the exchange step: result records -> all ranks (ordered on torch's stream after the handle's)
            be.pack_results(records.data_ptr(), torch.cuda.current_stream().cuda_stream)
            sharding.gather_records(records.cpu() if rehearsal else records, gathered, dist)

    for _ in range(args.warmup):
        step()
    be.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    be.sync()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = sharding.max_over_ranks(dt, dist, device="cpu" if rehearsal else f"cuda:{dev}")
    # what the collective really connected: every rank's (rank, device index), gathered over the same process group
    ranks_seen = None
    if world > 1:
        me = torch.tensor([[float(rank), float(dev)]], dtype=torch.float64, device="cpu" if rehearsal else f"cuda:{dev}")
        allr = torch.zeros((world, 2), dtype=torch.float64, device=me.device)
        sharding.gather_records(me, allr, dist)
        ranks_seen = [[int(a), int(b)] for a, b in allr.cpu().tolist()]
        g = gathered.cpu()
        ok = bool(torch.isfinite(g[:, -8]).all()) and sorted(int(x) for x in g[:, -2].tolist()) == sorted(
            i for r in range(world) for i in sharding.shard_window_ids(r, world, args.windows, None if args.scaling == "weak" else args.windows))
        if not ok:
            raise RuntimeError("the all-gathered result records do not cover every rank's windows")
    # ---- N > 1: BASELINE config 4 AS WRITTEN in the same run -- 1024 windows in total, block-partitioned over the ranks (1024 / 8 = 128
    #      per GPU), same step + exchange, its own handle (max_batch = the shard: the small-batch launch variants).  A second timed
    #      leg, reported as `strong_scaling_config4` beside the weak line, so that one driver run yields both.
    strong_leg = None
    if world > 1 and args.scaling == "weak" and 1024 % world == 0:
        ids_s = list(sharding.shard_window_ids(rank, world, 0, 1024))
        ws_s = synth.make_windows(ids_s, n_frames=args.frames, n_vo=args.vo, n_landmarks=args.landmarks)
        Ws = len(ws_s)
        be_s = backend.Backend(args.frames, args.vo, max_landmarks=args.landmarks, max_obs=max(w.n_obs for w in ws_s), max_batch=Ws)
        be_s.upload(ws_s)
        rec_s = torch.zeros((Ws, rec), dtype=torch.float64, device=f"cuda:{dev}")
        gat_s = torch.zeros((world * Ws, rec), dtype=torch.float64, device="cpu" if rehearsal else f"cuda:{dev}")

        def step_s():
            be_s.run_optimize(sync=False)
            be_s.pack_results(rec_s.data_ptr(), torch.cuda.current_stream().cuda_stream)
            sharding.gather_records(rec_s.cpu() if rehearsal else rec_s, gat_s, dist)
        steps_s = min(args.steps, 100)
        for _ in range(args.warmup):
            step_s()
        be_s.sync(); barrier()
        t0s = time.perf_counter()
        for _ in range(steps_s):
            step_s()
        be_s.sync(); barrier()
        dts = sharding.max_over_ranks(time.perf_counter() - t0s, dist, device="cpu" if rehearsal else f"cuda:{dev}")
        gs = gat_s.cpu()
        if not (bool(torch.isfinite(gs[:, -8]).all()) and sorted(int(x) for x in gs[:, -2].tolist()) == list(range(1024))):
            raise RuntimeError("strong leg: the all-gathered result records do not cover the 1024 windows")
        be_s.close()
        strong_leg = {"workload": f"BASELINE config 4 as written: 1024 windows in total, {Ws} per GPU on {world} GPUs, full backendOptimization() + one all-gather of the result records per step",
                      "scaling": "strong", "value": 1024 * steps_s / dts, "unit": "windows/s", "ms_per_step": 1e3 * dts / steps_s, "steps": steps_s, "windows_per_gpu": Ws, "n_gpus": world}
    # per-kernel-family HIP-event timing of one more (untimed, profiled) step, on the handle's own stream
    be.run_optimize(sync=True, profile=True)
    fam = be.last_timing(); cnt = be.last_counts()

    extra = {}
    t_incl = None
    if rank == 0 and world == 1 and not args.no_host_legs:
        def single_window_ms(N, Nvo, wins, mo):
            b1 = backend.Backend(N, Nvo, max_landmarks=args.landmarks, max_obs=mo, max_batch=1)
            b1.upload(wins[:1])
            ts1 = []
            for _ in range(9):                        # HIP events on the handle's stream (a host clock would add this process's
                b1.run_optimize(sync=True)            # stream-synchronise wake-up latency, which depends on the runtime's wait mode)
                ts1.append(float(b1.last_timing()[0]))
            b1.close()
            return float(np.median(ts1[2:]))
        # ms / optimize() of ONE window (BASELINE's second figure: the reference calls backendOptimization() once per frame)
        extra["ms_per_optimize_single_window"] = single_window_ms(args.frames, args.vo, windows, max_obs)

        # ---- BASELINE config 2: ONE 11-KF / 300-landmark window, residual + Jacobian kernels only, vs the CPU -----------
        b1 = backend.Backend(args.frames, args.vo, max_landmarks=args.landmarks, max_obs=max_obs, max_batch=1)
        b1.upload(windows[:1])
        tl = []
        for _ in range(25):
            b1.run_linearize(sync=True); tl.append(b1.last_timing()[:3].copy())
        tl = np.median(np.array(tl[5:]), axis=0)
        b1.close()
        F0, N = windows[0].n_factors, args.frames
        bytes_cfg2 = 284.0 * F0 + 6016.0 * (N - 1)              # algorithmic bytes of one window-linearisation
        cfg2 = {"workload": f"one synthetic window, N={N} KF, L={windows[0].L} landmarks, F={F0} reprojection + {N - 1} IMU factors + priors: every residual block evaluated once (ceres::Problem::Evaluate equivalent), isv_batch_linearize",
                "gpu_us": 1e3 * float(tl[0]), "gpu_us_proj_kernel": 1e3 * float(tl[1]), "gpu_us_imu_prior_kernels": 1e3 * float(tl[2]),
                "algorithmic_bytes": bytes_cfg2, "gpu_GBps": bytes_cfg2 / (float(tl[0]) * 1e-3) / 1e9,
                "note": "a single window moves 0.4 MB: launch-latency bound, far from the HBM roofline (SURVEY 8d); the batched figure is the k_lin_gram entry of roofline_by_kernel"}
        if cpu_lib is not None:
            dp = C.POINTER(C.c_double)
            ps = np.zeros((F0, 28)); im = np.zeros((N - 1, 465)); pr = np.zeros(256); co = np.zeros(1)
            cw = windows[0].c()
            tc = []
            for _ in range(60):
                t1 = time.perf_counter()
                cpu_lib.isvo_linearize(C.byref(cfg0), C.byref(cw), ps.ctypes.data_as(dp), im.ctypes.data_as(dp), pr.ctypes.data_as(dp), co.ctypes.data_as(dp))
                tc.append(time.perf_counter() - t1)
            cfg2["cpu_us"] = 1e6 * float(np.median(tc[10:])); cfg2["cpu_GBps"] = bytes_cfg2 / float(np.median(tc[10:])) / 1e9
            cfg2["cpu"] = "oracle isvo_linearize, 1 thread, same host"
        extra["config2_single_window_linearize"] = cfg2

        # ---- config 4 as written: 1024 windows over 8 GPUs = 128 per GPU; the 1-GPU figure at B = 128 ---------------------
        nb = min(128, W)
        b128 = backend.Backend(args.frames, args.vo, max_landmarks=args.landmarks, max_obs=max_obs, max_batch=nb)
        b128.upload(windows[:nb])
        for _ in range(3):
            b128.run_optimize(sync=True)
        t1 = time.perf_counter()
        for _ in range(40):
            b128.run_optimize(sync=False)
        b128.sync()
        t128 = (time.perf_counter() - t1) / 40
        b128.close()
        extra["strong_scaling_shard"] = {"workload": f"{nb} windows on one GPU (BASELINE config 4 as written: 1024 windows / 8 GPUs)", "ms_per_step": 1e3 * t128,
                                         "value": nb / t128, "unit": "windows/s"}

        # ---- the reference's compile-time shape: ALL_BUF_SIZE = 18, Vo_SIZE = 8 (include/parameters.h:35-40) ------------
        n18 = min(W, 1024)
        w18 = synth.make_windows(range(n18), n_frames=18, n_vo=8, n_landmarks=args.landmarks)
        mo18 = max(w.n_obs for w in w18)
        b18 = backend.Backend(18, 8, max_landmarks=args.landmarks, max_obs=mo18, max_batch=n18)
        b18.upload(w18)
        for _ in range(2):
            b18.run_optimize(sync=True)
        t1 = time.perf_counter()
        for _ in range(20):
            b18.run_optimize(sync=False)
        b18.sync()
        t18 = (time.perf_counter() - t1) / 20
        b18.run_optimize(sync=True, profile=True)
        fam18 = b18.last_timing(); cnt18 = b18.last_counts()
        b18.close()
        # one window per call on the handle the drop-in shim creates (include/isvins_estimator_shim.hpp: NUM_OF_F = 1000 landmarks,
        # NUM_OF_F x ALL_BUF_SIZE = 18 000 observations, Vo_SIZE + 1 roll/pitch slots, max_batch = 1): abi.make_config's defaults
        bshim = backend.Backend(18, 8)
        assert (bshim.cfg.max_landmarks, bshim.cfg.max_obs, bshim.cfg.max_batch) == (1000, 18000, 1)
        bshim.upload(w18[:1])
        ts1 = []
        for _ in range(9):
            bshim.run_optimize(sync=True); ts1.append(float(bshim.last_timing()[0]))
        cshim = bshim.last_counts()
        bshim.close()
        extra["reference_shape_n18_vo8"] = {"workload": f"{n18} windows, N=18 KF, Nvo=8, L={args.landmarks}, full backendOptimization()", "ms_per_step": 1e3 * t18,
                                            "value": n18 / t18, "unit": "windows/s",
                                            "build_solve_avg_launch_us": 1e3 * float(fam18[4]) / max(int(cnt18[1]), 1),
                                            "ms_per_optimize_single_window": float(np.median(ts1[2:])),
                                            "single_window_handle": "as the shim creates it: max_landmarks 1000, max_obs 18000, max_batch 1",
                                            "single_window_fused_lin_gram": int(cshim[4]) == 1, "single_window_solve_kernel": "k_build_solve_st" if int(cshim[6]) == 1 else "k_build_solve_sb",
                                            "ms_per_optimize_single_window_tight_handle": single_window_ms(18, 8, w18, mo18)}

        # ---- BASELINE config 5: ONE stress window, 20 KF / 2000 landmarks / exactly 30 000 reprojection factors --------------
        w5 = synth.make_window(0, n_frames=20, n_vo=8, n_landmarks=2000, target_factors=30000)
        b5 = backend.Backend(20, 8, max_landmarks=2000, max_obs=w5.n_obs, max_batch=1)
        b5.upload([w5])
        t5 = []
        for _ in range(12):
            b5.run_optimize(sync=True); t5.append(float(b5.last_timing()[0]))
        b5.run_optimize(sync=True, profile=True)
        fam5 = b5.last_timing(); cnt5 = b5.last_counts()
        b5.close()
        ms5 = float(np.median(t5[2:]))
        wi5 = max(int(cnt5[3]), 1); nl5 = max(int(cnt5[2]), 1)
        kl = np.diff(w5.lm_obs_ptr[: w5.L + 1]).astype(float)                       # track lengths k_l
        r1_dense = 2.0 * (6 * 20) ** 2 * w5.L; r1_sparse = float(np.sum(2.0 * (6.0 * kl) ** 2))
        # Only `window_iterations` of the launches of the elimination / solve kernels do work (a rejected step re-runs the dogleg
        # alone and the gated launches return at once, ~4 us each): per ACTIVE launch = family time / window_iterations (the
        # no-op launches' few us stay in the numerator: the figure errs on the slow side).  VERDICT r3 weak 4: the r03 line divided
        # by all launches and overstated frac 2x.
        split5 = int(cnt5[7])
        act = lambda fam_ms: 1e3 * float(fam_ms) / wi5
        if split5 > 0:
            dom5, r1_us = "k_schur_split<8,3>", act(fam5[2])
            kern5 = {"k_proj_linearize<0>": act(fam5[1]), "k_schur_split<8,3> (direct part + rank-1 downdates, one launch)": act(fam5[2]), "k_schur_fold": act(fam5[3])}
        else:
            dom5, r1_us = "k_rank1_mfma<8,3>", act(fam5[3])
            kern5 = {"k_proj_linearize<0>": act(fam5[1]), "k_sweep_mfma": act(fam5[2]), "k_rank1_mfma<8,3>": r1_us}
        kern5.update({"k_build_solve_sb<true,0>": act(fam5[4]), "k_dogleg (all 10 launches, mean)": 1e3 * float(fam5[5]) / 10.0,
                      "k_proj_linearize<1> + k_step_control (all 10 launches, mean)": 1e3 * float(fam5[6]) / 10.0})
        cfg5 = {"workload": f"one synthetic stress window: N=20 KF, Nvo=8, L={w5.L} landmarks, F={w5.n_factors} reprojection factors, full backendOptimization() (10 dogleg iterations + marginalisation)",
                "ms_per_optimize": ms5, "value": 1e3 / ms5, "unit": "windows/s", "window_iterations": wi5, "launches_per_family": nl5, "fused_lin_gram": int(cnt5[4]) == 1,
                "schur_split_groups": split5,
                "kernel_us_per_active_launch": kern5,
                "roofline": {"kernel": dom5, "bound": "mfma", "achieved": r1_dense / (r1_us * 1e-6) / 1e12 if r1_us > 0 else None,
                             "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": (r1_dense / (r1_us * 1e-6) / 1e12 / FP64_PEAK_TFLOPS) if r1_us > 0 else None,
                             "traffic": None, "active_launch_us": r1_us,
                             "flops_dense_equivalent": r1_dense, "flops_sparse_exact": r1_sparse,
                             "note": "the Schur panel contraction S -= sum_l c_l w_l w_l^T of ONE window, dense-equivalent 2 (6N)^2 L flops (SURVEY 8d), per ACTIVE launch; "
                                     + ("round 4: the window's landmarks are split over schur_split_groups workgroups (+ up to 16 for the direct part in the same launch) and folded in fixed order by k_schur_fold; MfmaUtil of the same launch: profiles/r05_config5_pmc_utilisation.csv"
                                        if split5 > 0 else "one workgroup of 12 wavefronts on one CU")}}
        if cpu_lib is not None:
            cfg5c = backend.abi.make_config(20, 8, max_landmarks=2000, max_obs=w5.n_obs, max_batch=1)
            tcs = []
            for _ in range(3):
                o5 = w5.clone(); s5 = backend.abi.isv_summary_t(); m5 = backend.abi.isv_marg_result_t()
                t1 = time.perf_counter(); cpu_lib.isvo_optimize(C.byref(cfg5c), C.byref(o5.c()), C.byref(s5), C.byref(m5)); tcs.append(time.perf_counter() - t1)
            cfg5["cpu_baseline"] = {"value": 1.0 / min(tcs), "unit": "windows/s", "ms_per_optimize": 1e3 * min(tcs), "cores": 1, "kind": "port",
                                    "sample": "the same window, oracle isvo_optimize (-O3 -march=native), best of 3"}
        extra["stress_config5"] = cfg5

        # ---- the pose-graph consumer: PoseGraph::optimizeCS passes, one graph and a batch ---------------
        try:
            from isvins_amd import posegraph as pgm
            K, loops, S = 200, 5, 1024
            graphs = [pgm.make_pose_graph(100 + s_, K, loops) for s_ in range(8)]
            def pgo_time(n_graphs, reps):
                opt = pgm.PoseGraphOptimizer(K, max_graphs=n_graphs, max_loop_blocks=8 * K)
                firsts = [graphs[s_ % 8][2] for s_ in range(n_graphs)]; curs = [K - 1] * n_graphs
                best, its, kms, nblk = None, 0, None, 0
                for rep in range(reps + 1):
                    batch = [pgm.clone_keyframes(graphs[s_ % 8][0]) for s_ in range(n_graphs)]
                    t1 = time.perf_counter(); res = opt.optimize_batch(batch, firsts, curs); dt = time.perf_counter() - t1
                    if rep > 0 and (best is None or dt < best):
                        best = dt; kms, nblk = opt.last_kernel_ms()
                    its = float(np.mean([r.iterations for r in res]))
                opt.close()
                return best, its, kms, nblk
            t_one, it_one, k_one, nb_one = pgo_time(1, 3)
            t_all, it_all, k_all, nb_all = pgo_time(S, 2)
            # algorithmic flops of one graph's pass: per LM iteration one factorisation of the 6x6-block skyline (a block row of
            # w blocks costs ~ w^2 block products of 2 * 6^3 flops, summed as nblk * mean w ~ 2 blocks for a chain + the loop
            # rows) and two triangular solves; then the selected inverse (~ 2 factorisations).  Priced per skyline BLOCK STEP:
            # 2 * 6^3 flops, the unit k_pgo's recurrences advance by.
            blk_steps = lambda nblk, its: nblk * (its + 1.0) * 3.0 + 2.0 * nblk * 3.0
            flops_one = 432.0 * blk_steps(nb_one, it_one); flops_all = 432.0 * blk_steps(nb_all, it_all)
            pgo = {"workload": f"PoseGraph::optimizeCS pass (LM <= 10 iterations + marginal covariances + write-back) over {K} keyframes with {loops} loop closures, synthetic",
                   "ms_one_graph": 1e3 * t_one, "lm_iterations_one_graph": it_one,
                   "batch_graphs": S, "ms_per_batch_call": 1e3 * t_all, "value": S / t_all, "unit": "graphs/s", "lm_iterations_batch_mean": it_all,
                   "what": "one isv_pgo_optimize_batch call: host structure analysis + H2D + k_pgo (a call that leaves CUs idle: four wavefronts per graph, k_pgo<4>; a batch that fills the GPU: one, k_pgo<1>; same bits) + D2H + write-back",
                   "kernel_ms_one_graph": k_one, "kernel_ms_batch": k_all,
                   "roofline": {"kernel": "k_pgo", "bound": "fp64-valu-latency", "achieved": flops_all / (k_all * 1e-3) / 1e12 if k_all else None, "peak": FP64_PEAK_TFLOPS,
                                "unit": "TFLOP/s", "frac": (flops_all / (k_all * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if k_all else None, "traffic": None,
                                "avg_launch_us": 1e3 * k_all if k_all else None, "skyline_blocks_per_graph": nb_all / S,
                                "block_steps_per_second_batch": blk_steps(nb_all, it_all) / (k_all * 1e-3) if k_all else None,
                                "us_per_block_step_one_graph": (1e3 * k_one / blk_steps(nb_one, it_one)) if k_one else None,
                                "note": "one wavefront per graph walks a chain of dependent 6x6 block steps (skyline Cholesky, Takahashi selected inverse): a latency chain, priced against the FP64 vector peak; the batch fills the GPU with graphs"}}
            if cpu_lib is not None and hasattr(cpu_lib, "isvo_pgo_optimize"):
                kfp = C.POINTER(pgm.isv_pg_keyframe_t)
                cpu_lib.isvo_pgo_optimize.argtypes = [C.POINTER(pgm.isv_pgo_config_t), C.c_int32, kfp, C.c_int32, C.c_int32, C.POINTER(pgm.isv_pgo_result_t)]
                cpu_lib.isvo_pgo_optimize.restype = C.c_int
                cfgp = pgm.make_config(K)
                def cpu_pgo(sparse, reps_):
                    cpu_lib.isvo_pgo_set_sparse(1 if sparse else 0)
                    best_ = None
                    try:
                        for _ in range(reps_):
                            o = pgm.clone_keyframes(graphs[0][0]); r = pgm.isv_pgo_result_t()
                            t1 = time.perf_counter(); cpu_lib.isvo_pgo_optimize(C.byref(cfgp), K, o, graphs[0][2], K - 1, C.byref(r)); dt_ = time.perf_counter() - t1
                            best_ = dt_ if best_ is None else min(best_, dt_)
                    finally:
                        cpu_lib.isvo_pgo_set_sparse(0)
                    return best_
                t_sky = cpu_pgo(True, 5); t_dense = cpu_pgo(False, 1)
                pgo["cpu_baseline"] = {"value": 1.0 / t_sky, "unit": "graphs/s", "cores": 1, "kind": "port", "ms_one_graph": 1e3 * t_sky,
                                       "sample": "one graph, best of 5: the oracle's LM loop on a SKYLINE (envelope) Cholesky with the covariance blocks by envelope solves (isvo_pgo_set_sparse; oracle/isv_pgo_oracle.c) -- the work a sparse direct solver such as the reference's SPARSE_NORMAL_CHOLESKY does on this matrix; -O3 -march=native, 1 thread (the reference runs this step in one background thread)",
                                       "ms_one_graph_dense_checker": 1e3 * t_dense}
            extra["pose_graph_optimisation"] = pgo
        except Exception as ex:                          # (secondary leg: never take the benchmark line down)
            extra["pose_graph_optimisation"] = {"error": repr(ex)}

        # ---- PCIe-inclusive rate of the RESIDENT path (never `value`): whole sequences through the native window manager with the
        #      windows kept on the device between frames -- per frame only the newest frame's observations, one IMU record and
        #      the propagated state go up, the newest / oldest poses and the solve_flags come back (tools/isv_replay, a child
        #      process: host feed + hand-over + GPU + read-back per lock-step frame) -- beside the same replay re-uploading every frame
        try:
            import tempfile
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import sequence_harness as _sh                      # (the stream simulator only: no oracle call)
            tool = os.path.join(ROOT, "tools", "isv_replay")
            with tempfile.TemporaryDirectory() as td:
                stream = os.path.join(td, "stream.txt")
                _sh.write_stream(stream, args.frames, args.vo, 48, seed=1)
                def replay(nseq, groups, extra):
                    o = subprocess.run([tool, stream, "--sequences", str(nseq), "--groups", str(groups), "--out", td, "--write", "0"] + extra, capture_output=True, text=True, timeout=300)
                    line = [l for l in o.stdout.splitlines() if l.startswith("{")]
                    return json.loads(line[-1]) if o.returncode == 0 and line else {"error": (o.stderr or o.stdout)[-300:]}
                r_res = replay(2048, 4, []); r_res8 = replay(4096, 8, []); r_up = replay(2048, 4, ["--no-resident"]); r_one = replay(1, 1, [])
            extra["device_resident_replay"] = {
                "workload": f"tools/isv_replay: a simulated 10 Hz camera / 200 Hz IMU stream (~280 landmarks and ~2300 reprojection factors per window: longer tracks than the synthetic benchmark windows) replayed into S sequences in lock step, every frame a full solveOdometry (triangulate + 10-iteration backendOptimization + marginalisation); K groups = K estimators on K host threads",
                "resident_2048_seq_4_groups_frames_per_s": r_res.get("frames_per_second"), "resident_4096_seq_8_groups_frames_per_s": r_res8.get("frames_per_second"),
                "reupload_2048_seq_4_groups_frames_per_s": r_up.get("frames_per_second"), "resident_one_sequence_frames_per_s": r_one.get("frames_per_second"),
                "mean_step_ms_resident_2048": r_res.get("mean_step_ms"), "errors": [r.get("error") for r in (r_res, r_res8, r_up, r_one) if "error" in r] or None,
                "unit": "frames/s"}
        except Exception as ex:
            extra["device_resident_replay"] = {"error": repr(ex)}

        # ---- PCIe-inclusive rate (never `value`): what a caller handing over HOST buffers sees --------------------------
        # (median of five hand-overs, each of a fresh copy of the windows: one cold hand-over varied by +-0.5 ms from run to run)
        legs = []
        for _ in range(5):
            w2 = [w.clone() for w in windows]           # download() writes into the windows: use a second copy
            ptrs = be.marshal(w2)                       # ctypes marshalling is the Python harness's cost, not the C ABI's
            t1 = time.perf_counter()
            be.upload(w2, ptrs=ptrs); t_up = time.perf_counter() - t1
            be.run_optimize(sync=True); t_opt = time.perf_counter() - t1 - t_up
            be.download(w2, ptrs=ptrs, as_list=False)
            legs.append((time.perf_counter() - t1, t_up, t_opt))
        t_incl, t_up, t_opt = sorted(legs)[len(legs) // 2]
        # the same hand-over, pipelined: two handles driven by two host threads (ctypes releases the GIL inside the
        # C calls), so one batch packs / copies while the other is on the GPU; 3 batches per handle
        import threading
        be2 = backend.Backend(args.frames, args.vo, max_landmarks=args.landmarks, max_obs=max_obs, max_batch=W)
        w3 = [w.clone() for w in windows]; ptrs3 = be2.marshal(w3)
        be2.upload(w3, ptrs=ptrs3); be2.run_optimize(sync=True)      # first-use warm-up of the second handle
        init2 = [w.clone() for w in windows]; init3 = [w.clone() for w in windows]
        pi2 = be.marshal(init2); pi3 = be2.marshal(init3)
        reps = 3

        def drive(b, ws_, p_):
            for _ in range(reps):
                b.upload(ws_, ptrs=p_); b.run_optimize(sync=True)
                b.download(w2 if b is be else w3, ptrs=ptrs if b is be else ptrs3, as_list=False)
        th = [threading.Thread(target=drive, args=(be, init2, pi2)), threading.Thread(target=drive, args=(be2, init3, pi3))]
        t1 = time.perf_counter()
        for t_ in th: t_.start()
        for t_ in th: t_.join()
        t_pipe = (time.perf_counter() - t1) / (2 * reps)
        be2.close()

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        value = world * W * args.steps / dt
        N, L = args.frames, args.landmarks
        Fw = Ftot / W
        Lw = sum(w.L for w in windows) / W
        n_lin, n_bs, n_sw = max(int(cnt[0]), 1), max(int(cnt[1]), 1), max(int(cnt[2]), 1)
        win_iters = max(int(cnt[3]), 1)                 # window-iterations that were linearised + solved (gated windows excluded)
        lin_ms, sw_ms, r1_ms, bs_ms, dg_ms, sc_ms = (float(fam[i]) for i in range(1, 7))
        # ALGORITHMIC work of one window-iteration
        tvis = (36 * N * (N + 1) // 2 + 18 * N) * 8.0
        bytes_lin = Fw * (24.0 + 280.0) + Lw * (32.0 + 104.0)      # in: factor record, observation (+ landmark depth / host point); out: 224 B strip, cost, 48 B w (+ landmark scalars, host w)
        bytes_sweep = Fw * (26 * 8.0 + 4.0) + tvis                 # [J_i | J_j | r] of every factor once + the permutation; out: packed pose blocks, gradient, diagonal
        bytes_rank1 = (Fw + Lw) * 48.0 + Lw * 20.0 + 2.0 * tvis    # packed w vectors, {c_l, g_l}, metadata; read-modify-write of the packed blocks
        # fused path (k_lin_gram + landmark prologue of k_rank1_mfma): no 224-B strips.  k_lin_gram: in = factor stream record (8 B) +
        # observation (16 B) per factor, depth + host point (32 B) per landmark; out = cost (8 B) + observer w (48 B) + landmark
        # pieces (64 B) per factor, the packed pose blocks per window.  k_rank1_mfma additionally reads the 64-B pieces and writes
        # the landmark scalars (56 B) and the host w (48 B).
        fused = int(cnt[4]) == 1
        bytes_lingram = Fw * (24.0 + 120.0) + Lw * 32.0 + tvis
        if fused:
            bytes_rank1 += Fw * 64.0 + Lw * 104.0
        # k_dogleg: back-substitution from the packed w vectors (48 B / observation) + landmark scalars in (5) / out (4)
        # + tangent vectors + candidate states + IMU / prior J^T J blocks for the model cost
        bytes_dogleg = (Fw + Lw) * 48.0 + Lw * 72.0 + 15 * N * 8.0 * 8 + 2 * 16 * N * 8.0 + (N - 1) * (495 + 64 + 225) * 8.0
        # k_step_control: factor record + observation + landmark depth (x and candidate) per factor, states
        bytes_control = Fw * 24.0 + Lw * (24.0 + 24.0) + 3 * 16 * N * 8.0
        flops_bs = bs_flops(N)
        pmc = {}
        for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                break
            except Exception:
                pmc = {}

        def roof(kernel, bound, work_per_winiter, ms_sum, launches, peak, unit, scale, note=None):
            # achieved = algorithmic work of all launches / their total duration (gated windows do no work)
            ach = work_per_winiter * win_iters / (ms_sum * 1e-3) / scale if ms_sum > 0 else None
            t = pmc.get(kernel, {}).get("hbm_bytes_per_launch") if pmc.get("windows_per_gpu") == W else None
            r = {"kernel": kernel, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                 "frac": (ach / peak) if ach else None, "traffic": t,
                 "avg_launch_us": ms_sum / launches * 1e3 if ms_sum > 0 else None, "launches_per_step": launches,
                 "algorithmic_work_per_window_iteration": work_per_winiter}
            if note:
                r["note"] = note
            return r

        solve_kernel = "k_build_solve_st" if int(cnt[6]) == 1 else "k_build_solve_sb"
        roofs = [roof(solve_kernel, "fp64-valu-latency", flops_bs, bs_ms, n_bs, FP64_PEAK_TFLOPS, "TFLOP/s", 1e12,
                      ("k_build_solve_st (round 4): 256 threads and <= 40 KB of LDS per window, four windows per CU; the speed/bias chain blocks stream through small LDS rings, the pose-block downdates are FP64 MFMA tiles. " if int(cnt[6]) == 1 else "")
                      + "chains of dependent 6x6 / 9x9 FP64 pivots (v_readlane -> rsq -> Newton) in LDS: latency bound; priced against the FP64 vector peak, which equals the FP64 matrix peak on MI355X"),
                 roof("k_rank1_mfma", "hbm", bytes_rank1, r1_ms, n_sw, HBM_PEAK_GBS, "GB/s", 1e9)]
        if fused:
            roofs.insert(1, roof("k_lin_gram", "hbm", bytes_lingram, lin_ms, n_lin, HBM_PEAK_GBS, "GB/s", 1e9,
                                 "ProjectionFactor::Evaluate fused with the Gram products: the Jacobian strips stay on the CU; lane per factor + FP64 MFMA, latency / issue bound at 3 workgroups per CU"))
        else:
            roofs.insert(1, roof("k_proj_linearize<0>", "hbm", bytes_lin, lin_ms, n_lin, HBM_PEAK_GBS, "GB/s", 1e9))
            roofs.insert(2, roof("k_sweep_mfma", "hbm", bytes_sweep, sw_ms, n_sw, HBM_PEAK_GBS, "GB/s", 1e9))
        fused_control = int(cnt[5]) == 1
        if not fused_control:
            roofs.append(roof("k_dogleg", "hbm", bytes_dogleg, dg_ms, args_iters(be), HBM_PEAK_GBS, "GB/s", 1e9))
            roofs.append(roof("k_step_control", "hbm", bytes_control, sc_ms, args_iters(be), HBM_PEAK_GBS, "GB/s", 1e9))
        else:                  # k_dogleg<true>: the candidate evaluation and the step control run in the same kernel
            roofs.append(roof("k_dogleg", "hbm", bytes_dogleg + bytes_control, dg_ms + sc_ms, args_iters(be), HBM_PEAK_GBS, "GB/s", 1e9,
                              "k_dogleg<true>: back-substitution, dogleg step, candidate costs and TrustRegionMinimizer step control in one kernel"))
        sums = {solve_kernel: bs_ms, "k_proj_linearize<0>": lin_ms, "k_lin_gram": lin_ms, "k_sweep_mfma": sw_ms, "k_rank1_mfma": r1_ms, "k_dogleg": dg_ms + (sc_ms if fused_control else 0), "k_step_control": sc_ms}
        dominant = max(roofs, key=lambda r: sums[r["kernel"]])
        total_w = world * W
        out = {
            "metric": "sliding-window solves/sec (11 KF, ~300 landmarks)", "value": value, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "ms_per_optimize_batched": ms_step / W,
            "ms_per_optimize_single_window": extra.get("ms_per_optimize_single_window"),      # one resident window on a second handle, HIP events
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{W} independent synthetic sliding windows per GPU (N={N} KF, Nvo={args.vo}, L={L} landmarks, F~{Fw:.0f} reprojection factors each; BASELINE config 4 batch of config-2 windows), full backendOptimization(): NUM_ITERATIONS=10 dogleg iterations + update() + double2vector + MargForward/MargBackward on every window"
                                   + ("" if world == 1 else f"; + one RCCL all-gather of the {total_w} per-window result records ({rec * 8} B each) per step"),
                       "windows_per_gpu": W, "windows_total": total_w, "frames": N, "landmarks": L, "factors_total": Ftot, "iterations_cap": 10,
                       "parallelism": f"independent windows block-partitioned over {world} rank(s) ({args.scaling} scaling), no collective inside the solve"
                                      + ("" if world == 1 else "; exchange step = ncclAllGather of the result records on device buffers"),
                       "ranks_seen_by_the_collective": ranks_seen, "collective_backend": None if world == 1 else ("gloo (one-GPU rehearsal)" if rehearsal else "nccl (RCCL)")},
            "roofline": dominant, "roofline_by_kernel": roofs,
            "host_buffers_inclusive": None if t_incl is None else {
                "value": W / t_incl, "unit": "windows/s", "ms_per_batch": 1e3 * t_incl,
                "pipelined_two_handles": {"value": W / t_pipe, "unit": "windows/s", "ms_per_batch": 1e3 * t_pipe},
                "ms_upload": 1e3 * t_up, "ms_optimize": 1e3 * t_opt, "ms_download": 1e3 * (t_incl - t_up - t_opt),
                "what": "one isv_batch_upload (raw CSR into ONE pinned block on min(16, allowed CPUs) host threads + ONE H2D copy + k_upload_build: the solver's view -- pair groups, schedule, factor stream -- is derived on the device, round 5) + isv_batch_optimize + isv_batch_download (D2H) of the same batch, pageable host buffers.  This is the STATELESS hand-over (every call carries whole windows); consecutive frames of the same sequences keep their windows on the device instead: device_resident_replay",
                "resident_path_frames_per_s_2048_sequences": (extra.get("device_resident_replay") or {}).get("resident_2048_seq_4_groups_frames_per_s")},
            "kernel_ms": {"profiled_step_total_events": float(fam[0]), "lin_gram_or_proj_linearize_sum": lin_ms, "sweep_mfma_sum": sw_ms, "rank1_mfma_sum": r1_ms, "build_solve_sum": bs_ms,
                          "dogleg_sum": dg_ms, "step_control_sum": sc_ms, "window_iterations": win_iters},
            "cpu_baseline": cpu,
            # SURVEY 8d / BASELINE.md 3 promise a stock-Ceres cross-check "if find_package(Ceres) succeeds on the GPU box": probed in
            # round 3 (`find / -name ceres.h -o -path '*eigen3/Eigen/Dense' -o -name so3.hpp`: nothing; gpurun_out/r3_probe) -- the
            # MI355X image has no Eigen, Ceres or Sophus either, so the oracle stays unpinned by the reference
            "ceres": "unavailable",
        }
        if strong_leg is not None:
            out["strong_scaling_config4"] = strong_leg
        for k in ("config2_single_window_linearize", "strong_scaling_shard", "reference_shape_n18_vo8", "stress_config5", "device_resident_replay", "pose_graph_optimisation"):
            if k in extra:
                out[k] = extra[k]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def args_iters(be):
    return max(int(be.cfg.num_iterations), 1)


if __name__ == "__main__":
    main()
