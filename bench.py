#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sampleblock -> BCS -> BCM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Workload (config.workload): BASELINE.json configs[1] -- the reference demo's shape on one GPU:
2.5 Msps x 20 ms windows (S=50000), 8 SVs, rngrid3-format random ENU-dt grid of 25^4 = 390625 points
plus the velocity-drift grid of the same size, synthetic I/Q (the demo recording is not shipped).
One "step" = one pass of the hot path over a batch of `--windows` windows resident in HBM:
BatchCorrScores (DC sum, lag/Doppler banks, finalize) + BatchCorrManifold (pos scan, vel scan,
fused arg-max) for every window, each with its own channel state.

metric = manifold gridpoints x SVs correlated per second (both manifolds), whole job.
N>1: the grid dimension is sharded (each rank scores its own contiguous slice of an N-times larger
global grid -> weak scaling), stage 1 is recomputed per rank, and the per-window arg-max is
exchanged with one RCCL all-reduce(MAX) of packed (score,index) keys per step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def pmc_traffic(windows):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json; FETCH_SIZE doubled as the microarch guide prescribes), or None when
    the batch size differs from the profiled one."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        d = json.load(open(f))
        if d.get("windows_per_step") == windows:
            k = d["kernels"].get("bcm_scan_kernel")
            return k["hbm_bytes_per_launch"] if k else None
    return None


def cpu_baseline(cfg, budget_s=10.0, mp_budget_s=8.0):
    """Oracle (fp64 port of the reference algorithm: FFT BatchCorrScores in numpy + C grid scan) timed on the host,
    on whole windows of the same workload: one thread (the contract's cpu_baseline), then one process per usable
    core over independent windows, with the host description beside them."""
    import shutil
    import subprocess
    import tempfile
    import navlab_dpe_sdr_amd as dpe
    from oracle import mp_baseline as mb
    fs, S, K, G, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["G"], cfg["L"], cfg["B"]
    iq, cs, ce, bw = dpe.workload.build_windows(2, fs, S, K, seed=99, amp=cfg["amp"])
    _, _, pos, vel, _ = dpe.workload.build_grids(G)
    d = {"fs": np.float64(fs), "S": np.int64(S), "L": np.int64(L), "B": np.int64(B), "iq": iq, "cs": cs, "ce": ce,
         "bw": bw, "pos": pos, "vel": vel}
    mb.o.lib()
    n, t0 = 0, time.perf_counter()
    while True:
        mb.full_window(d, n % 2)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    out = {"value": n * 2.0 * G * K / dt, "unit": "gridpoint*SV/s", "cores": 1, "kind": "port",
           "sample": "%d full windows (FFT BCS in numpy + C grid scan, fp64), %.1f s" % (n, dt),
           "x_realtime": n / dt / 50.0, "host": mb.host_info()}
    # all usable cores: independent processes (no profiler preload, one thread each) behind a file barrier
    # at most 64 workers: the windows stream 67 MB FFT batches, and on the 256-thread host of the GPU box 256
    # workers measured half the aggregate rate of 64 (memory bound), 14 s per window
    cores = min(len(os.sched_getaffinity(0)), 64)
    rundir = tempfile.mkdtemp(prefix="dpe_cpu_baseline_")
    try:
        mb.save_workload(os.path.join(rundir, "workload.npz"), **d)
        env = {k: v for k, v in os.environ.items()
               if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "OMPI_", "RANK", "LOCAL_RANK"))}
        env.update(OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=ROOT)
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.mp_baseline", rundir, str(i), str(mp_budget_s)],
                                  cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                 for i in range(cores)]
        t_wait = time.time()
        while sum(os.path.exists(os.path.join(rundir, "ready.%d" % i)) for i in range(cores)) < cores:
            if time.time() - t_wait > 120 or any(p.poll() is not None for p in procs):
                break
            time.sleep(0.02)
        open(os.path.join(rundir, "go"), "w").close()
        res = []
        for p in procs:
            try:
                so, _ = p.communicate(timeout=mp_budget_s + 60)
                res.append(json.loads(so.decode().strip().splitlines()[-1]))
            except Exception:
                p.kill()
        if len(res) == cores:
            tot, span = sum(r["n"] for r in res), max(r["dt"] for r in res)
            out["all_cores"] = {"value": tot * 2.0 * G * K / span, "unit": "gridpoint*SV/s", "cores": cores,
                                "sample": "%d windows over %d single-thread processes, %.1f s" % (tot, cores, span),
                                "x_realtime": tot / span / 50.0}
        else:
            out["all_cores"] = {"value": None, "cores": cores, "sample": "only %d of %d workers reported" % (len(res), cores)}
    finally:
        shutil.rmtree(rundir, ignore_errors=True)
    return out


def acq_main(mode):
    """`bench.py --acq MODE`: coarse-acquisition timing on one MI355X (BASELINE.json configs[4] shape): 32 PRNs x
    125 Doppler bins x all 2500 code delays of a 10 ms / 2.5 Msps window.  Prints one JSON line: search cells per second
    (PRN x bin x delay), ms per window, and -- as its cpu_baseline leg -- the oracle's numpy fp64 restatement of the
    reference's coarse_acquisition on a bounded sample.  Not the headline metric."""
    import torch
    import navlab_dpe_sdr_amd as dpe
    fs, S = 2.5e6, 25000
    ch = dpe.synth.random_channels(77, 6, prns=[3, 7, 11, 18, 22, 31])
    ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(78, fs, S, ch, amp=150.0, flip=np.zeros(6, dtype=bool))
    bins = np.arange(-62, 63) * 100.0
    prns = list(range(1, 33))
    acq = dpe.Acquisition(fs, S, prns, bins, mode=mode, prn_chunk=32)
    d = torch.from_numpy(iq).to("cuda:0")
    for _ in range(3):
        acq.search(d)
    torch.cuda.synchronize()
    t = dpe.engine.HipEventTimer()
    n = 20
    t.start()
    for _ in range(n):
        acq.search(d)
    t.stop()
    ms = t.elapsed_ms() / n
    res = acq.results()
    acq.search_signal(d)
    t0 = time.perf_counter()
    for _ in range(5):
        full = acq.search_signal(d)                    # coarse + fine frequency, host-synchronous
    ms_full = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    for _ in range(5):
        acq.search(d); coarse = acq.results()
    ms_coarse = (time.perf_counter() - t0) / 5 * 1e3
    cells = len(prns) * bins.size * (S // 10)
    out = {"metric": "acquisition search cells (PRN x Doppler bin x code delay) per second", "mode": mode,
           "value": cells / (ms * 1e-3), "ms_per_window": ms, "x_realtime": 10.0 / ms,
           "search_signal_ms_per_window": ms_full, "search_plus_results_ms": ms_coarse,
           "found": sorted(r["prn"] for r in res if r["found"]), "truth": sorted(int(p) for p in ch["prn"])}
    from oracle import oracle as o
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 8.0:
        o.coarse_acquisition(iq, fs, prns[k % 32], bins, coherent=(mode == "coherent"), mode="textbook" if mode == "textbook" else None)
        k += 1
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": k * bins.size * (S // 10) / dt, "cores": 1, "kind": "port", "sample": "%d PRNs, %.1f s" % (k, dt)}
    print(json.dumps(out))



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=None, help="windows per step (batch resident in HBM); default 256 (R) / 32 (H)")
    ap.add_argument("--config", choices=["R", "H"], default="R",
                    help="R = BASELINE.json configs[1] (the metric's configuration); H = configs[2] (25 Msps, 12 SVs, 1e5-point grids)")
    ap.add_argument("--exchange", choices=["keys", "scores"], default="keys",
                    help="multi-GPU exchange: packed arg-max keys (8 B/window/manifold) or the north-star-literal "
                         "all-reduce(SUM) of the zero-initialised full score vectors")
    ap.add_argument("--acq", choices=["coherent", "noncoherent", "textbook"], default=None,
                    help="time the cold-start acquisition search instead (8f row 4; separate JSON line, not the headline)")
    ap.add_argument("--clock-warmup-s", type=float, default=1.0,
                    help="seconds of untimed steps after --warmup, before the timed region (GPU clock settle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scores", action="store_true", help="skip the per-point score write (arg-max only)")
    ap.add_argument("--include-h2d", action="store_true",
                    help="also time the steps with the window batch uploaded from pinned host memory inside the timed "
                         "region (reported as pcie_inclusive_value; never the headline value)")
    args = ap.parse_args()
    if args.acq:
        return acq_main(args.acq)

    import torch
    import torch.distributed as dist
    import navlab_dpe_sdr_amd as dpe

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("DPE_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank RCCL self-test
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # DPE_BENCH_BACKEND=gloo DPE_BENCH_SHARE_GPU=1: functional check of the N > 1 code path on a one-GPU box (all
        # ranks on cuda:0, exchange through gloo) -- RCCL refuses two ranks on one device.  Never a performance number.
        backend = os.environ.get("DPE_BENCH_BACKEND", "nccl")
        if os.environ.get("DPE_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if use_dist else 0)

    cfg = dict(dpe.workload.CONFIG_R if args.config == "R" else dpe.workload.CONFIG_H)
    fs, S, K, G, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["G"], cfg["L"], cfg["B"]
    W = args.windows if args.windows else (256 if args.config == "R" else 32)
    iq, cs, ce, bw = dpe.workload.build_windows(W, fs, S, K, seed=0, amp=cfg["amp"])
    pos_g, vel_g, pos, vel, off = dpe.workload.build_grids(G, rank, world)
    write_scores = (not args.no_scores) or args.exchange == "scores"

    iq_d = torch.from_numpy(iq).to(dev)          # inputs resident in HBM before the timed region
    bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=W,
                                max_channels=K, write_scores=write_scores, pos_index_offset=off, vel_index_offset=off)
    bcm.Start()
    stream = torch.cuda.current_stream()
    key_views = {}

    def keys_tensor():
        """torch view of the handle's packed keys of the LAST Update (two device sets alternate); int64:
        scores are >= 0 so the sign bit is clear"""
        ptr = bcm.Keys
        if ptr not in key_views:
            class _Cai:
                __cuda_array_interface__ = {"shape": (W, 2), "typestr": "<i8", "data": (ptr, False), "version": 2}
            key_views[ptr] = torch.as_tensor(_Cai(), device=dev)
        return key_views[ptr]

    if use_dist:
        if args.exchange == "scores":
            glob_p = torch.zeros((W, G * world), dtype=torch.float32, device=dev)
            glob_v = torch.zeros((W, G * world), dtype=torch.float32, device=dev)

            class _CaiS:
                def __init__(self, ptr):
                    self.__cuda_array_interface__ = {"shape": (W, G), "typestr": "<f4", "data": (ptr, False), "version": 2}
            loc_p = torch.as_tensor(_CaiS(bcm.PosScores), device=dev)
            loc_v = torch.as_tensor(_CaiS(bcm.VelScores), device=dev)

    pending = [None]   # the previous step's arg-max exchange, still in flight

    def drain():
        if pending[0] is not None:
            pending[0].wait()            # stream-level wait: the compute stream continues after the collective
            pending[0] = None

    def step():
        bcs.Update(iq_d, cs, stream=stream)     # stage 1 overlaps the previous step's exchange
        drain()                                 # ... which must be done before the scan clears that key set
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce, stream=stream)
        if use_dist:
            if args.exchange == "keys":
                # in-order exchange by default; DPE_BENCH_EXCH_MODE=async overlaps it with the next step's stage 1
                # (measured SLOWER on one MI355X: the collective's kernel then competes with the correlator)
                mode = os.environ.get("DPE_BENCH_EXCH_MODE", "sync")
                if mode == "async":
                    _, pending[0] = dpe.sharding.allreduce_argmax(keys_tensor(), dist, async_op=True)
                elif mode == "sync":
                    dpe.sharding.allreduce_argmax(keys_tensor(), dist)
            else:
                glob_p.zero_(); glob_v.zero_()
                glob_p[:, off:off + G].copy_(loc_p); glob_v[:, off:off + G].copy_(loc_v)
                dist.all_reduce(glob_p, op=dist.ReduceOp.SUM)
                dist.all_reduce(glob_v, op=dist.ReduceOp.SUM)
                torch.argmax(glob_p, dim=1); torch.argmax(glob_v, dim=1)

    def fence():
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Clock warm-up, untimed: a GPU that idled at its lowest sclk takes on the order of a second of load to settle at its
    # sustained clock -- the first bench of a fresh box measured 3 % below an immediate second one with only the
    # --warmup steps in front.  The metric is steady-state throughput, so the settle time stays outside the timed region.
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.clock_warmup_s:
        for _ in range(32):
            step()
        fence()
    # Untimed side pass: per-kernel HIP events on every kernel, for `kernels_ms_per_step` (informational).  It runs
    # BEFORE the timed region so that it also serves as clock warm-up (the first ~20 launches of a fresh process run
    # 5-10 % slower; rocprofv3 per-launch trace, profiles/README.md).
    extra = 24   # ~22 ms: enough for the clocks to settle whatever --warmup / --steps are
    bcs.profile(True); bcm.profile(True)
    for _ in range(extra):
        step()
    fence()
    kern_extra = bcs.profile(False)
    bcm.profile(False)
    # Timed region: only the dominant kernel (the fused scan, `roofline`) carries HIP events -- a pair of events
    # around every kernel costs ~2 % of the step (measured: 0.948 vs 0.928 ms).
    bcm.profile(True)
    if os.environ.get("DPE_BENCH_EVENTS") == "all":   # experiment: events around every kernel inside the timed region
        bcs.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kern = {}
    kern.update(bcm.profile(False))
    if os.environ.get("DPE_BENCH_EVENTS") == "all":
        bcs.profile(False)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # result sanity on rank 0: the synthetic windows put the truth at the grid centre -> the ML point must be
    # the grid point with the smallest geometric offset pattern; just check the fix is finite and in-grid.
    if use_dist:
        res = bcm.results_from_keys(keys_tensor().cpu().numpy().view(np.uint64), pos_g, vel_g)
        if world == 1:   # self-test: the exchanged keys decode to what the handle itself reports
            ref = bcm.results()
            assert all(a["posIndex"] == b["posIndex"] and a["velIndex"] == b["velIndex"] and
                       np.array_equal(a["zVal"], b["zVal"]) for a, b in zip(res, ref))
            res = ref
    else:
        res = bcm.results()
    assert all(np.isfinite(r["zVal"]).all() for r in res)
    if world == 1:   # the banks must cover every index the grids reach
        assert all(r["posOutOfWindow"] == 0 and r["velOutOfWindow"] == 0 for r in res), "bank window too narrow"

    pcie_value = pcie_overlapped = None
    if args.include_h2d and not use_dist:
        iq_pin = torch.from_numpy(iq).pin_memory()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            iq_d.copy_(iq_pin, non_blocking=True)     # SampleBlock's H2D leg (sampleblock.cu:356-410), same stream
            step()
        fence()
        pcie_value = float(args.steps) * W * 2.0 * G * K / (time.perf_counter() - t1)
        # the same with the upload double-buffered on a copy stream (what SampleBlock does per window): batch n+1
        # travels while batch n is processed
        copy_stream = torch.cuda.Stream()
        bufs = [iq_d, torch.empty_like(iq_d)]
        ready = [torch.cuda.Event(), torch.cuda.Event()]
        freed = [torch.cuda.Event(), torch.cuda.Event()]
        for e in freed:
            e.record(torch.cuda.current_stream())

        def upload(i):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[i])
                bufs[i].copy_(iq_pin, non_blocking=True)
                ready[i].record(copy_stream)

        fence()
        t2 = time.perf_counter()
        upload(0)
        for n in range(args.steps):
            cur = n & 1
            if n + 1 < args.steps:
                upload(cur ^ 1)
            torch.cuda.current_stream().wait_event(ready[cur])
            bcs.Update(bufs[cur], cs, stream=stream)
            bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce, stream=stream)
            freed[cur].record(torch.cuda.current_stream())
        fence()
        pcie_overlapped = float(args.steps) * W * 2.0 * G * K / (time.perf_counter() - t2)

    if rank == 0:
        units = float(args.steps) * W * 2.0 * G * K * world       # (gridpoint, SV) pairs, both manifolds, all ranks
        value = units / dt
        windows_per_s = args.steps * W / dt
        ms_scan, n_scan = kern["bcm_scan"]
        # one launch scans BOTH manifolds: 16 B grid read + 4 B score write per point
        bytes_per_launch = 2 * W * 20.0 * G
        ach = bytes_per_launch / (ms_scan / n_scan * 1e-3) / 1e9 if n_scan else 0.0
        out = {
            "metric": "manifold gridpoints x SVs correlated/sec", "value": value, "unit": "gridpoint*SV/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"], "samples_per_window": S, "svs": K, "grid_points_per_manifold_per_gpu": G,
                       "manifolds": 2, "windows_per_step": W, "lag_half_width": L, "bin_half_width": B,
                       "exchange": args.exchange if use_dist else "none", "scores_written": write_scores},
            "x_realtime": windows_per_s / 50.0, "windows_per_s": windows_per_s,
            "roofline": {"bound": "hbm", "kernel": "bcm_scan_kernel", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(W) if world == 1 else None,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": ms_scan / n_scan if n_scan else None},
            "kernels_ms_per_step": dict({k: v[0] / args.steps for k, v in kern.items()},
                                        **{k: v[0] / max(v[1], 1) for k, v in kern_extra.items()}),
        }
        if pcie_value is not None:
            out["pcie_inclusive_value"] = pcie_value
            out["pcie_inclusive_overlapped_value"] = pcie_overlapped
        if world == 1:
            # measured ceiling beside the nominal peak: stream copy and triad over 1 GiB arrays
            cp, tr = dpe.engine.hbm_ceiling(1 << 30, 10, stream)
            tb = out["roofline"]["traffic"]
            out["roofline"]["measured_ceiling"] = {
                "copy_GBps": cp, "triad_GBps": tr, "algorithmic_over_triad": ach / tr,
                # what the kernel really pulls from HBM (PMC bytes / launch time): the grids are shared by the windows
                # of a batch and stay in L2, so the algorithmic rate may exceed the physical ceiling
                "physical_GBps": tb / (ms_scan / n_scan * 1e-3) / 1e9 if tb and n_scan else None}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
    bcm.Stop(); bcs.Stop()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes a version banner through C stdio on stdout; push it out first so that the JSON line is the
    # last thing this process prints
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
