#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X sliding-window VIO backend.

Metric (BASELINE.json): sliding-window solves/sec on synthetic 11-KF / ~300-landmark windows, plus
ms per optimize().  One STEP = one pass of the hot path (Estimator::backendOptimization():
vector2double + problemSolve [<= 10 dogleg iterations] + update() + double2vector [+ marginalisation
where built]) over the rank's device-resident batch of windows.  Inputs are resident in HBM before the
timed region (the step restores the pristine uploaded state on device first).

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one process per GPU with torch.distributed.run (RCCL over xGMI);
windows are independent, so ranks shard them with no data-path collective ("weak" scaling:
--windows per rank) and only the timing is max-reduced.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import isvins_loader  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector = matrix peak


def cpu_baseline(windows, cfg, budget_s=12.0):
    """the CPU oracle (oracle/, a 1-thread C port of the reference algorithm) timed on a bounded sample"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from isvins_amd import abi
    lib = oracle_lib.load()
    n = 0
    t0 = time.perf_counter()
    for w in windows:
        o = w.clone()
        s = abi.isv_summary_t(); mg = abi.isv_marg_result_t()
        lib.isvo_optimize(C.byref(cfg), C.byref(o.c()), C.byref(s), C.byref(mg))
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "windows/s", "cores": 1, "kind": "port",
            "sample": f"{n} of the benchmark's own 11-KF/300-landmark windows, full backendOptimization(), oracle/libisv_oracle.so (gcc -O2), 1 thread",
            "ms_per_optimize": 1e3 * dt / n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=1024, help="windows per GPU (BASELINE config 4 batch)")
    ap.add_argument("--frames", type=int, default=11)
    ap.add_argument("--vo", type=int, default=5)
    ap.add_argument("--landmarks", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-legs", action="store_true", help="skip the untimed host-buffer (PCIe-inclusive) legs; used for the rocprofv3 passes so that their kernel statistics hold the benchmark's own launches only")
    args = ap.parse_args()

    # more than two handles' worth of HIP streams in one process (the untimed two-handle leg below): the runtime maps
    # streams onto 4 hardware queues by default; must be set before the runtime initialises.  One handle is unaffected.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ISV_BENCH_REHEARSAL=1: rehearse the multi-rank path on ONE GPU (every rank on cuda:0, gloo instead of RCCL, which
    # refuses two ranks on one device); the driver's real runs have one GPU per rank
    rehearsal = os.environ.get("ISV_BENCH_REHEARSAL") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)
    isvins_loader.load()
    from isvins_amd import backend, synth
    backend.build()

    # The C-ABI library allocates on the calling thread's current HIP device.  backend.load_library() maps torch's
    # HIP runtime first, so the library and torch share ONE runtime and torch.cuda.set_device() above selects the
    # device for both; the free-memory check below verifies that the batch really landed on this rank's GPU.
    dev = local_rank if world > 1 else 0
    free_before = torch.cuda.mem_get_info(dev)[0]

    W = args.windows
    from isvins_amd import sharding
    ids = sharding.shard_window_ids(rank, world, W)
    windows = synth.make_windows(ids, n_frames=args.frames, n_vo=args.vo, n_landmarks=args.landmarks)
    Ftot = sum(w.n_factors for w in windows)
    max_obs = max(w.n_obs for w in windows)
    be = backend.Backend(args.frames, args.vo, max_landmarks=args.landmarks, max_obs=max_obs, max_batch=W)
    be.upload(windows)
    free_after = torch.cuda.mem_get_info(dev)[0]
    if W >= 64 and free_before - free_after < (64 << 20):
        raise RuntimeError(f"rank {rank}: the backend did not allocate on cuda:{dev} (free memory moved by {free_before - free_after} B)")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        be.run_optimize(sync=True)
    barrier()
    t0 = time.perf_counter()
    fam = np.zeros(8); cnt = np.zeros(8)
    for _ in range(args.steps):
        be.run_optimize(sync=False)
    be.sync()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = sharding.max_over_ranks(dt, dist, device="cpu" if rehearsal else "cuda")
    # per-kernel-family HIP-event timing of one more (untimed, profiled) step, on the handle's own stream
    be.run_optimize(sync=True, profile=True)
    fam = be.last_timing(); cnt = be.last_counts()
    # PCIe-inclusive rate (never `value`): what a caller handing over HOST buffers sees, isv_batch_upload (host packing +
    # H2D) -> isv_batch_optimize -> isv_batch_download (D2H + unpack into the Estimator arrays)
    # ms / optimize() of ONE window (BASELINE's second figure: the reference calls backendOptimization() once per frame):
    # a handle of its own with one resident window, median of 7 solves
    ms_single = None
    if rank == 0 and world == 1 and not args.no_host_legs:
        be1 = backend.Backend(args.frames, args.vo, max_landmarks=args.landmarks, max_obs=max_obs, max_batch=1)
        be1.upload(windows[:1])
        ts1 = []
        for _ in range(9):                        # HIP events on the handle's stream (a host clock would add this process's
            be1.run_optimize(sync=True)           # stream-synchronise wake-up latency, which depends on the runtime's wait mode)
            ts1.append(float(be1.last_timing()[0]))
        ms_single = float(np.median(ts1[2:]))
        be1.close()
    t_incl = None
    if rank == 0 and world == 1 and not args.no_host_legs:
        w2 = [w.clone() for w in windows]               # download() writes into the windows: use a second copy
        ptrs = be.marshal(w2)                           # ctypes marshalling is the Python harness's cost, not the C ABI's
        t1 = time.perf_counter()
        be.upload(w2, ptrs=ptrs); t_up = time.perf_counter() - t1
        be.run_optimize(sync=True); t_opt = time.perf_counter() - t1 - t_up
        be.download(w2, ptrs=ptrs, as_list=False)
        t_incl = time.perf_counter() - t1
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
        lin_ms, sw_ms, r1_ms, bs_ms = float(fam[1]), float(fam[2]), float(fam[3]), float(fam[4])
        # ALGORITHMIC work of one window-iteration
        tvis = (36 * N * (N + 1) // 2 + 18 * N) * 8.0
        bytes_lin = Fw * (24.0 + 280.0) + Lw * (32.0 + 104.0)      # in: factor record, observation (+ landmark depth / host point); out: 224 B strip, cost, 48 B w (+ landmark scalars, host w)
        bytes_sweep = Fw * (26 * 8.0 + 4.0) + tvis                 # [J_i | J_j | r] of every factor once + the permutation; out: packed pose blocks, gradient, diagonal
        bytes_rank1 = (Fw + Lw) * 48.0 + Lw * 20.0 + 2.0 * tvis    # packed w vectors, {c_l, g_l}, metadata; read-modify-write of the packed blocks

        def bs_flops(N):
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
        flops_bs = bs_flops(N)
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        except Exception:
            pmc = {}

        def roof(kernel, bound, work_per_winiter, ms_sum, launches, peak, unit, scale):
            # achieved = algorithmic work of all launches / their total duration (gated windows do no work)
            ach = work_per_winiter * win_iters / (ms_sum * 1e-3) / scale if ms_sum > 0 else None
            t = pmc.get(kernel, {}).get("hbm_bytes_per_launch") if pmc.get("windows_per_gpu") == W else None
            return {"kernel": kernel, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                    "frac": (ach / peak) if ach else None, "traffic": t,
                    "avg_launch_us": ms_sum / launches * 1e3 if ms_sum > 0 else None, "launches_per_step": launches,
                    "algorithmic_work_per_window_iteration": work_per_winiter}

        roofs = [roof("k_build_solve_sb", "mfma", flops_bs, bs_ms, n_bs, FP64_PEAK_TFLOPS, "TFLOP/s", 1e12),
                 roof("k_proj_linearize<0>", "hbm", bytes_lin, lin_ms, n_lin, HBM_PEAK_GBS, "GB/s", 1e9),
                 roof("k_sweep_mfma", "hbm", bytes_sweep, sw_ms, n_sw, HBM_PEAK_GBS, "GB/s", 1e9),
                 roof("k_rank1_mfma", "hbm", bytes_rank1, r1_ms, n_sw, HBM_PEAK_GBS, "GB/s", 1e9)]
        sums = {"k_build_solve_sb": bs_ms, "k_proj_linearize<0>": lin_ms, "k_sweep_mfma": sw_ms, "k_rank1_mfma": r1_ms}
        dominant = max(roofs, key=lambda r: sums[r["kernel"]])
        out = {
            "metric": "sliding-window solves/sec (11 KF, ~300 landmarks)", "value": value, "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "ms_per_optimize_batched": ms_step / W,
            "ms_per_optimize_single_window": ms_single,      # one resident window on a second handle, HIP events
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{W} independent synthetic sliding windows per GPU (N={N} KF, Nvo={args.vo}, L={L} landmarks, F~{Fw:.0f} reprojection factors each; BASELINE config 4 batch of config-2 windows), full backendOptimization(): NUM_ITERATIONS=10 dogleg iterations + update() + double2vector + MargForward/MargBackward on every window",
                       "windows_per_gpu": W, "frames": N, "landmarks": L, "factors_total": Ftot, "iterations_cap": 10,
                       "parallelism": f"independent windows sharded over {world} rank(s), no data-path collective"},
            "roofline": dominant, "roofline_by_kernel": roofs,
            "host_buffers_inclusive": None if t_incl is None else {
                "value": W / t_incl, "unit": "windows/s", "ms_per_batch": 1e3 * t_incl,
                "pipelined_two_handles": {"value": W / t_pipe, "unit": "windows/s", "ms_per_batch": 1e3 * t_pipe},
                "ms_upload": 1e3 * t_up, "ms_optimize": 1e3 * t_opt, "ms_download": 1e3 * (t_incl - t_up - t_opt),
                "what": "one isv_batch_upload (host packing + H2D) + isv_batch_optimize + isv_batch_download (D2H) of the same batch, pageable host buffers; packing on min(8, cores) host threads"},
            "kernel_ms": {"profiled_step_total_events": float(fam[0]), "proj_linearize_sum": lin_ms, "sweep_mfma_sum": sw_ms, "rank1_mfma_sum": r1_ms, "build_solve_sum": bs_ms, "window_iterations": win_iters},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(windows, be.cfg)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
