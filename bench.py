#!/usr/bin/env python3
"""bench.py -- benchmark of the sampleblock -> BCS -> BCM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a torch.distributed environment: this process starts
        `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child and relays its output.

Headline workload (config.workload, the LAST JSON line): BASELINE.json configs[1] -- the reference demo's shape on one GPU:
2.5 Msps x 20 ms windows (S = 50000), 8 SVs, rngrid3-format random ENU-dt grid of 25^4 = 390625 points plus the
velocity-drift grid of the same size, synthetic I/Q (the demo recording is not shipped).  One "step" = one pass of the hot
path over a batch of `--windows` windows resident in HBM: BatchCorrScores (DC sum, lag/Doppler banks, finalize) +
BatchCorrManifold (pos scan, vel scan, fused arg-max) for every window, each with its own channel state.
metric = manifold gridpoints x SVs correlated per second (both manifolds), whole job.

Without --config the run first prints one line each for the other configurations BASELINE.json names (same format,
"headline": false):  acq (configs[4]: cold-start acquisition search, N = 1 only), H (configs[2]: 25 Msps, 12 SVs, 1e5-point
grids; N = 1 only) and M (configs[3]: 1e6-point GLOBAL grids
sharded over the N GPUs, strong scaling), then the headline line.

N > 1: the grid dimension is sharded (R, H: each rank scores its own contiguous slice of an N-times larger global grid ->
weak scaling; M: the global grid is fixed -> strong scaling).  Stage 1 is sharded by WINDOW: rank r correlates windows
[r W/N, (r+1) W/N) and the banks travel with one RCCL all-gather each (code, carrier; K (2L+1 + 2B+1) 8 B per window);
`--stage1 replicated` recomputes them on every rank instead.  The per-window arg-max is exchanged with one RCCL
all-reduce(MAX) of packed (score, index) keys per step (`--exchange scores`: the north-star-literal all-reduce(SUM) of
zero-initialised full score vectors).

Timing: W untimed warm-up steps and a clock warm-up, then FIVE timed batches, each bracketed by barrier + synchronize and
each made of a whole number of repetitions of the K steps so that it lasts >= 0.2 s; per batch the MAX over ranks, and the
MEDIAN batch is reported.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector (non-matrix) peak, 256 CUs x 128 lanes x 2 flop x 2.4 GHz

# What physically limits each kernel (rocprofv3 SQ counters, profiles/*_sq_counters.json): none of them
# is HBM-bound -- the "hbm" roofline below is SURVEY 8(d)'s algorithmic-bytes convention, not the physical limiter.
BOUND_PHYSICAL = {"bcm_scan_kernel": "valu", "bcs_bank_chip_kernel": "valu+lds", "bcs_bank_chip2_kernel": "valu+lds", "bcs_bank16_kernel": "valu",
                  "bcs_bank_wide_kernel": "valu", "bcs_bank_kernel": "valu", "bcs_finalize_kernel": "latency",
                  "bcs_sum_kernel": "hbm"}


def pmc_traffic(config, windows, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json; FETCH_SIZE
    doubled as the microarch guide prescribes) for this configuration and batch size -> (bytes, source file) or (None, None)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        d = json.load(open(f))
        if d.get("config", "R") != config or d.get("windows_per_step") != windows:
            continue
        for k, v in d["kernels"].items():
            if kernel in k or kernel in v.get("full_name", ""):
                return v["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    return None, None


def cpu_baseline(cfg, budget_s=10.0, mp_budget_s=8.0, all_cores=True, windows=None):
    """Oracle (fp64 port of the reference algorithm: FFT BatchCorrScores in numpy + C grid scan) timed on the host,
    on whole windows of the same workload: one thread (the contract's cpu_baseline), then -- `all_cores`, the headline
    only -- one process per usable core over independent windows, with the host description beside them.
    `windows`: (iq, cs, ce, bw) of the GPU line's own synthetic batch, of which the first two are timed (H: 1.7 s of numpy
    to synthesise each, so they are not built twice); at least one whole window is always completed (H: one window is
    12 SVs x (five 500 000-point + one 4 194 304-point fp64 transforms) + the two scans, several seconds)."""
    import shutil
    import tempfile
    import navlab_dpe_sdr_amd as dpe
    from oracle import mp_baseline as mb
    fs, S, K, G, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["G"], cfg["L"], cfg["B"]
    if windows is not None:
        iq, cs, ce, bw = (np.ascontiguousarray(a[:2]) for a in windows)
    else:
        iq, cs, ce, bw = dpe.workload.build_windows(2, fs, S, K, seed=99, amp=cfg["amp"])
    _, _, pos, vel, _ = dpe.workload.build_grids(G)
    d = {"fs": np.float64(fs), "S": np.int64(S), "L": np.int64(L), "B": np.int64(B), "iq": iq, "cs": cs, "ce": ce,
         "bw": bw, "pos": pos, "vel": vel}
    mb.o.lib()
    n, t0 = 0, time.perf_counter()
    while True:
        mb.full_window(d, n % iq.shape[0])
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    out = {"value": n * 2.0 * G * K / dt, "unit": "gridpoint*SV/s", "cores": 1, "kind": "port",
           "sample": "%d full windows (FFT BCS in numpy + C grid scan, fp64), %.1f s" % (n, dt),
           "x_realtime": n / dt / 50.0, "host": mb.host_info()}
    if not all_cores:
        return out
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


def acq_line(modes=("coherent", "textbook", "noncoherent"), cpu_budget_s=5.0):
    """Cold-start acquisition (BASELINE.json configs[4]) as a measured line: 32 PRNs x 125 Doppler bins x all
    2500 code delays of a 10 ms / 2.5 Msps window, the whole search (wipe-off + fold, forward rocFFT, spectrum product + inverse
    transforms + |.| surface + per-delay maximum -- one fused kernel in the coherent mode --, peak statistics) timed between HIP
    events on its stream.
    `value` = search cells (PRN x bin x delay) per second of the reference's coherent semantics (fixture O8); `modes` holds the
    textbook "1 ms coherent x 10 non-coherent" form too (BASELINE's wording; not a reference algorithm) and the reference's
    non-coherent mode (coherent = False: one 25 000-point correlation per bin, |.| over the ten lag aliases) -- on the same 125-bin
    raster, and on the raster the reference itself uses for it (25 bins x 500 Hz, correlator.py:13).  Roofline: the
    search is memory-bound by construction -- its batched transforms are 6e8 flop per window against 40 MB of surface -- so the
    stanza prices the algorithmic bytes (samples read once + the |.| surface written once) against the HBM peak and quotes
    the transform flop rate beside it.  cpu_baseline: the oracle's numpy restatement of coarse_acquisition on a bounded sample."""
    import torch
    import navlab_dpe_sdr_amd as dpe
    fs, S, N = 2.5e6, 25000, 10
    M = S // N
    ch = dpe.synth.random_channels(77, 6, prns=[3, 7, 11, 18, 22, 31])
    ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(78, fs, S, ch, amp=150.0, flip=np.zeros(6, dtype=bool))
    bins = np.arange(-62, 63) * 100.0
    prns = list(range(1, 33))
    d = torch.from_numpy(iq).to("cuda:0")
    cells = len(prns) * bins.size * M
    per_mode = {}
    found = None
    for mode in modes:
        acq = dpe.Acquisition(fs, S, prns, bins, mode=mode, prn_chunk=32)
        for _ in range(5):
            acq.search(d)
        torch.cuda.synchronize()
        t = dpe.engine.HipEventTimer()
        n = 50
        t.start()
        for _ in range(n):
            acq.search(d)
        t.stop()
        ms = t.elapsed_ms() / n
        res = acq.results()
        acq.search_signal(d)                        # (the first call plans the fine-frequency transform)
        t0 = time.perf_counter()
        for _ in range(5):
            acq.search_signal(d)                    # coarse + statistics + fine frequency, host-synchronous
        ms_full = (time.perf_counter() - t0) / 5 * 1e3
        per_mode[mode] = {"ms_per_window": ms, "cells_per_s": cells / (ms * 1e-3), "search_signal_ms_per_window": ms_full}
        if mode == modes[0]:
            found = sorted(r["prn"] for r in res if r["found"])
            assert found == sorted(int(p) for p in ch["prn"]), "acquisition did not find the simulated PRNs"
        del acq
        if mode == "noncoherent":     # ... and on DOPPLER_SEARCH_MATRIX_NONCOHERENT, the raster the reference pairs with this mode
            b25 = np.arange(-12, 13) * 500.0
            acq = dpe.Acquisition(fs, S, prns, b25, mode=mode, prn_chunk=32)
            for _ in range(5):
                acq.search(d)
            torch.cuda.synchronize()
            t = dpe.engine.HipEventTimer()
            t.start()
            for _ in range(n):
                acq.search(d)
            t.stop()
            ms25 = t.elapsed_ms() / n
            per_mode["noncoherent_25x500Hz"] = {"ms_per_window": ms25, "cells_per_s": len(prns) * b25.size * M / (ms25 * 1e-3)}
            del acq
    ms0 = per_mode[modes[0]]["ms_per_window"]
    alg_bytes = 4.0 * S + 4.0 * cells                    # int16 I/Q once + the fp32 |.| surface once
    n_fft = bins.size + len(prns) * bins.size            # forward (time-folded rows) + inverse transforms of length M
    flops = n_fft * 5.0 * M * math.log2(M) + 6.0 * cells
    # Transform flops of each mode (5 n log2 n per complex transform + 6 per spectrum product), priced against the fp32 VECTOR peak:
    # the transform kernels are LDS / VALU bound, so this -- not the HBM fraction -- says how far each is from its limiter.
    mode_flops = {"coherent": flops,
                  "textbook": N * n_fft * 5.0 * M * math.log2(M) + 6.0 * N * cells,           # ten 2500-point correlations per (PRN, bin)
                  "noncoherent": n_fft * 5.0 * S * math.log2(S) + 6.0 * N * cells}             # one 25 000-point correlation per (PRN, bin)
    mode_flops["noncoherent_25x500Hz"] = (25 + len(prns) * 25) * 5.0 * S * math.log2(S) + 6.0 * N * len(prns) * 25 * M
    for m_, d_ in per_mode.items():
        d_["transform_flops"] = mode_flops[m_]
        d_["TFLOPs"] = mode_flops[m_] / (d_["ms_per_window"] * 1e-3) / 1e12
        d_["flop_frac"] = d_["TFLOPs"] / FP32_VECTOR_PEAK_TFLOPS
    out = {"metric": "acquisition search cells (PRN x Doppler bin x code delay) per second", "value": cells / (ms0 * 1e-3),
           "unit": "cell/s", "n_gpus": 1, "ms_per_step": ms0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "headline": False,
           "config": {"workload": "acq: cold-start acquisition, 32 PRNs x 125 Doppler bins (100 Hz) x 2500 code delays, 10 ms at 2.5 Msps "
                                  "(BASELINE.json configs[4])", "mode": modes[0], "prns": len(prns), "bins": int(bins.size), "delays": M},
           "x_realtime": 10.0 / ms0, "modes": per_mode, "found": found,
           "roofline": {"bound": "hbm", "bound_physical": "the fused transform kernel's VALU / LDS work and the one-block-per-PRN statistics chain",
                        "kernel": "dpe_acq_search (coherent / textbook: acq_wipe[_fold] + forward transform + acq_corr2500 [product, 2500-point "
                                  "inverse transforms, |.| summed over the code periods in the textbook mode, column max]; non-coherent: "
                                  "acq_fwd25k_pack + acq_corr25k_pack [ten packed 2500-point transforms per (PRN, bin), ten-point stage, alias sums]; "
                                  "+ acq_stats_small)",
                        "achieved": alg_bytes / (ms0 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / (ms0 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ms0,
                        "transform_flops_per_window": flops, "achieved_TFLOPs": flops / (ms0 * 1e-3) / 1e12,
                        "flop_peak_TFLOPs": FP32_VECTOR_PEAK_TFLOPS, "flop_frac": flops / (ms0 * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS}}
    from oracle import oracle as o
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < cpu_budget_s:
        o.coarse_acquisition(iq, fs, prns[k % 32], bins, coherent=True)
        k += 1
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": k * bins.size * M / dt, "unit": "cell/s", "cores": 1, "kind": "port",
                           "sample": "%d PRNs x 125 bins (numpy restatement of coarse_acquisition, fp64), %.1f s" % (k, dt)}
    return out


def acq_main(mode):
    """`bench.py --acq MODE`: the acquisition line alone, for the chosen mode first."""
    modes = (mode,) + tuple(m for m in ("coherent", "textbook", "noncoherent") if m != mode)
    print(json.dumps(acq_line(modes=modes, cpu_budget_s=8.0)))


class Ctx:
    """Process-group state of this rank."""
    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.use_dist = self.world > 1 or os.environ.get("DPE_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank RCCL self-test
        self.comm = None        # --comm dpe: the dpe_comm handle that carries the data-path exchange
        self.backend = os.environ.get("DPE_BENCH_BACKEND", "nccl")
        self.dist = None
        self.dev = None


def device_view(ptr, shape, typestr, dev):
    """torch tensor over device memory owned by a library handle."""
    import torch

    class _Cai:
        __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Cai(), device=dev)


def run_workload(name, ctx, args, steps, warmup, headline, cached=None):
    """Times one configuration; rank 0 returns the result dict (None elsewhere).  `cached`: windows built for another
    configuration of the same sampling shape (R and M share theirs)."""
    import torch
    import navlab_dpe_sdr_amd as dpe
    dist, dev, world, rank = ctx.dist, ctx.dev, ctx.world, ctx.rank
    cfg = dict({"R": dpe.workload.CONFIG_R, "H": dpe.workload.CONFIG_H, "M": dpe.workload.CONFIG_M}[name])
    fs, S, K, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["L"], cfg["B"]
    strong = name == "M"
    W = args.windows if (args.windows and headline) else (128 if name == "H" else 256)
    if not headline and args.extra_windows:
        W = args.extra_windows
    # --- inputs
    distinct = W
    if cached is not None and cached["key"] == (fs, S, K, W):
        iq, cs, ce, bw = cached["data"]
    else:
        distinct = W if name != "H" else min(W, 8)      # H: 8 distinct windows (1.7 s of numpy each), repeated with their state
        iq, cs, ce, bw = dpe.workload.build_windows(distinct, fs, S, K, seed=0, amp=cfg["amp"])
        if distinct < W:
            rep = (W + distinct - 1) // distinct
            iq, cs, ce, bw = (np.concatenate([a] * rep)[:W] for a in (iq, cs, ce, bw))
    if strong:
        G_global = cfg["G"]
        pos_g, vel_g, pos, vel, off = dpe.workload.build_grids_strong(G_global, rank, world)
    else:
        pos_g, vel_g, pos, vel, off = dpe.workload.build_grids(cfg["G"], rank, world)
        G_global = cfg["G"] * world
    G = pos.shape[0]
    write_scores = (not args.no_scores) or args.exchange == "scores"
    # --- stage-1 sharding by window
    # (the 1-rank RCCL self-test keeps the all-gather in the loop: a gather over one rank is a copy through the same calls)
    if ctx.use_dist and args.stage1 == "sharded" and W % world != 0:
        # (no silent fallback to replicated stage 1: the line would describe another exchange than the one asked for)
        if rank == 0:
            sys.stderr.write("bench.py: --stage1 sharded needs the %d windows of a step to divide over the %d ranks "
                             "(use --windows / --extra-windows, or --stage1 replicated)\n" % (W, world))
        sys.exit(2)
    shard1 = ctx.use_dist and args.stage1 == "sharded"
    Wl = W // world if shard1 else W
    w0 = rank * Wl if shard1 else 0
    iq_d = torch.from_numpy(np.ascontiguousarray(iq[w0:w0 + Wl])).to(dev)     # inputs resident in HBM before the timed region
    cs_l = np.ascontiguousarray(cs[w0:w0 + Wl])
    # The library's batches-in-flight form (include/dpe_hip.h, dpe_pipe_*): `--in-flight` lanes -- a BatchCorrScores /
    # BatchCorrManifold handle pair and a stream each, one device copy of the grids -- that consecutive steps are dealt to, so that
    # stage 1 of step n + 1 runs beside the grid scan of step n (what the reference gets from SampleBlock's ring and its side
    # streams).  The isolated-kernel passes and `one_stream_ms_per_step` run with one lane (dpe_pipe_set_in_flight(1)).
    n_lanes = max(1, args.in_flight)
    pipe = dpe.Pipe(fs, S, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=Wl, max_channels=K, in_flight=n_lanes,
                    write_scores=write_scores, pos_index_offset=off, vel_index_offset=off, bcm_max_windows=W)
    stream = torch.cuda.current_stream()
    nLag, nBin = 2 * L + 1, 2 * B + 1
    lanes = []           # per lane: handles, stream, and the buffers of this rank's exchanges
    for i in range(n_lanes):
        b_, m_, ls = pipe.lane_at(i)
        ln = {"bcs": b_, "bcm": m_, "stream": ls, "code": b_.CodeScores, "carr": b_.CarrScores, "key_views": {}}
        if ctx.use_dist:
            ln["tstream"] = torch.cuda.ExternalStream(ls, device=dev)      # torch.distributed enqueues on torch's CURRENT stream
        if shard1:
            ln["loc_code"] = device_view(b_.CodeScores, (Wl, K, nLag, 2), "<f4", dev)
            ln["loc_carr"] = device_view(b_.CarrScores, (Wl, K, nBin, 2), "<f4", dev)
            ln["full_code"] = torch.empty((W, K, nLag, 2), dtype=torch.float32, device=dev)
            ln["full_carr"] = torch.empty((W, K, nBin, 2), dtype=torch.float32, device=dev)
            ln["code"], ln["carr"] = ln["full_code"].data_ptr(), ln["full_carr"].data_ptr()
        if ctx.use_dist and args.exchange == "scores":
            ln["glob_p"] = torch.zeros((W, G_global), dtype=torch.float32, device=dev)
            ln["glob_v"] = torch.zeros((W, G_global), dtype=torch.float32, device=dev)
            ln["loc_p"] = device_view(m_.PosScores, (W, m_.PosScoresPitch), "<f4", dev)[:, :G]     # rows are 128-byte aligned (pitch >= G)
            ln["loc_v"] = device_view(m_.VelScores, (W, m_.VelScoresPitch), "<f4", dev)[:, :G]
        lanes.append(ln)
    by_handle = {ln["bcs"]._h.value: ln for ln in lanes}
    bcs, bcm = lanes[0]["bcs"], lanes[0]["bcm"]          # lane 0: the one-stream passes
    last = [None]        # lane of the last step

    def keys_tensor(ln):
        """torch view of a lane's packed keys of ITS last Update (two device sets alternate per handle); int64:
        scores are >= 0 so the sign bit is clear"""
        ptr = ln["bcm"].Keys
        if ptr not in ln["key_views"]:
            ln["key_views"][ptr] = device_view(ptr, (W, 2), "<i8", dev)
        return ln["key_views"][ptr]

    def gather_banks(ln):
        if ctx.comm is not None:      # --comm dpe: the C-ABI's own exchange (dpe_bcs_allgather_banks)
            ln["bcs"].allgather_banks(ctx.comm, ln["code"], ln["carr"], stream=ln["stream"])
        elif ctx.backend == "nccl":
            dist.all_gather_into_tensor(ln["full_code"], ln["loc_code"])
            dist.all_gather_into_tensor(ln["full_carr"], ln["loc_carr"])
        else:   # gloo (functional tests on one GPU): through host memory
            for full, loc in ((ln["full_code"], ln["loc_code"]), (ln["full_carr"], ln["loc_carr"])):
                parts = [torch.empty(loc.shape, dtype=loc.dtype) for _ in range(world)]
                dist.all_gather(parts, loc.cpu())
                full.copy_(torch.cat(parts))

    def step(src=None, in_stream=None):
        src = iq_d if src is None else src
        if not ctx.use_dist:           # one GPU: the whole step is one C call (dpe_pipe_submit)
            # (the timed region's samples are resident in HBM since before it started: no input dependency -- DPE_STREAM_NONE;
            #  the PCIe-inclusive passes hand over the stream their upload runs on)
            t = pipe.submit(src, cs_l, bw, ce, stream=in_stream or dpe.engine.STREAM_NONE)
            last[0] = (t, lanes[0])
            return
        # N > 1: the same lane, driven in pieces -- this rank's exchanges go between the stages, on the lane's stream
        t, b_, m_, ls = pipe.acquire(in_stream or dpe.engine.STREAM_NONE)
        ln = by_handle[b_._h.value]
        b_.Update(src, cs_l, stream=ls)
        pipe.mark_stage1(t)
        with torch.cuda.stream(ln["tstream"]):
            if shard1:
                gather_banks(ln)
            m_.Update(ln["code"], ln["carr"], bw, ce, stream=ls)
            if ctx.comm is not None:  # --comm dpe: dpe_bcm_exchange_keys, all-reduce(MAX) in place on the handle's keys
                m_.exchange_keys(ctx.comm, stream=ls, to_host=False)
            elif args.exchange == "keys":
                dpe.sharding.allreduce_argmax(keys_tensor(ln), dist)
            else:
                ln["glob_p"].zero_(); ln["glob_v"].zero_()
                ln["glob_p"][:, off:off + G].copy_(ln["loc_p"]); ln["glob_v"][:, off:off + G].copy_(ln["loc_v"])
                dist.all_reduce(ln["glob_p"], op=dist.ReduceOp.SUM)
                dist.all_reduce(ln["glob_v"], op=dist.ReduceOp.SUM)
                torch.argmax(ln["glob_p"], dim=1); torch.argmax(ln["glob_v"], dim=1)
        pipe.commit(t, W, K)
        last[0] = (t, ln)

    def fence():
        if ctx.use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if not ctx.use_dist:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device=dev if ctx.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(n_batches, n_steps):
        out_ms = []
        for _ in range(n_batches):
            fence()
            t0 = time.perf_counter()
            for _ in range(n_steps):
                step()
            fence()
            out_ms.append(max_over_ranks(time.perf_counter() - t0) / n_steps * 1e3)
        return out_ms

    pipe.set_in_flight(1)
    for _ in range(warmup):
        step()
    fence()
    # Clock warm-up, untimed: a GPU that idled at its lowest sclk takes on the order of a second of load to settle at its
    # sustained clock.  The metric is steady-state throughput, so the settle time stays outside the timed region.
    t_w = time.perf_counter()
    while max_over_ranks(time.perf_counter() - t_w) < args.clock_warmup_s:
        for _ in range(8):
            step()
        fence()
    # Untimed side pass: HIP events around every kernel (`kernels_ms_per_step`), which also names the time-dominant kernel.
    bcs.profile(True); bcm.profile(True)
    n_side = 12
    for _ in range(n_side):
        step()
    fence()
    side = dict(bcs.profile(False))
    side.update(bcm.profile(False))
    kernels_ms = {k: v[0] / max(v[1], 1) * (v[1] / n_side) for k, v in side.items()}   # per step (side chunks of a wide lag window add up)
    dominant = max(kernels_ms, key=kernels_ms.get)
    est = timed(1, steps)[0] * 1e-3
    reps = max(1, int(math.ceil(args.min_batch_s / max(est * steps, 1e-9))))
    # One stream (one lane): the isolated dominant kernel between HIP events for the `roofline` stanza (a pair of events around
    # every kernel costs ~2 % of a step, so only the dominant one carries them), and `one_stream_ms_per_step`.
    if dominant == "bcm_scan":
        bcm.profile(True)
    else:
        bcs.profile(dominant)
    one_ms = timed(args.batches if n_lanes == 1 else 3, reps * steps)
    kern = dict(bcm.profile(False)) if dominant == "bcm_scan" else dict(bcs.profile(False))
    one_stream_ms = float(np.median(one_ms))
    # The timed region proper: `--in-flight` lanes
    if n_lanes > 1:
        pipe.set_in_flight(n_lanes)
        for _ in range(2 * n_lanes):
            step()
        batch_ms = timed(args.batches, reps * steps)
    else:
        batch_ms = one_ms
    ms_step = float(np.median(batch_ms))
    # Stage-1 device status after the timed region: batches of the high-rate chip kernel compute their DC sums inside the stage-1
    # launch, and a correlator block that had to wait too long for them sums its window itself (right, but slower) -- bits 2 / 16
    # say that it happened; a timed region in which it did is not the steady state this line claims.
    stage1_status = 0
    have_status = False
    for ln in lanes:
        if ln["bcs"].stage1_kernel != "bcs_bank_chip2_kernel":
            continue      # (no device-side status on the other paths: per-sample kernels with host parameters)
        try:
            stage1_status |= int(ln["bcs"].dev_status(stream=ln["stream"]))
            have_status = True
        except dpe.engine.DpeError:
            pass          # (a batch too small for the riding DC sums)
    if not have_status:
        stage1_status = None
    # bits 1 / 2 / 8 are input errors (PRN / code frequency out of range, a broken chip-kernel promise): the line would describe
    # another computation -- fail loudly (an exception, not an assert: `python -O` must not strip it).  Bits 4 / 16 (a correlator
    # block summed its window itself) leave the results exact: the status travels in the JSON line as `stage1_dev_status`.
    if stage1_status and stage1_status & 11:
        raise RuntimeError("stage-1 device status %r after the timed region" % stage1_status)

    # result sanity on rank 0: finite fix; every lane's last batch gives the same answer (same inputs); with one rank the
    # exchanged keys decode to what the handle itself reports, and the banks must cover every index the grids reach
    t_last, ln_last = last[0]
    if ctx.use_dist:
        with torch.cuda.stream(ln_last["tstream"]):
            kt = keys_tensor(ln_last).cpu()
        res = ln_last["bcm"].results_from_keys(kt.numpy().view(np.uint64), pos_g, vel_g)
        if world == 1:
            ref = pipe.results(t_last)
            assert all(a["posIndex"] == b["posIndex"] and a["velIndex"] == b["velIndex"] and
                       np.array_equal(a["zVal"], b["zVal"]) for a, b in zip(res, ref))
            res = ref
    else:
        res = pipe.results(t_last)
        if n_lanes > 1:
            r2 = pipe.results(t_last - 1)
            # (DPE_BENCH_TIMING_ONLY: A/B builds under scratch/ab whose results are wrong on purpose -- scripts/ab_lib.sh, never the product)
            assert os.environ.get("DPE_BENCH_TIMING_ONLY") or all(
                a["posIndex"] == b_["posIndex"] and a["velIndex"] == b_["velIndex"] and np.array_equal(a["zVal"], b_["zVal"])
                for a, b_ in zip(r2, res)), "the lanes disagree"
    assert all(np.isfinite(r["zVal"]).all() for r in res)
    fixes = [[int(r["posIndex"]), int(r["velIndex"]), float(r["posScore"]), float(r["velScore"])] for r in res[:16]]
    if world == 1:
        assert all(r["posOutOfWindow"] == 0 and r["velOutOfWindow"] == 0 for r in res), "bank window too narrow"

    pcie_value = pcie_overlapped = None
    if args.include_h2d and headline and not ctx.use_dist:
        iq_pin = torch.from_numpy(iq).pin_memory()
        fence()
        t1 = time.perf_counter()
        for _ in range(steps):
            pipe.samples_consumed(last[0][0], stream=stream)     # the refill waits (on the device) for stage 1 of the step before
            iq_d.copy_(iq_pin, non_blocking=True)                # SampleBlock's H2D leg (sampleblock.cu:356-410), same stream
            step(in_stream=stream)
        fence()
        pcie_value = float(steps) * W * 2.0 * G * K / (time.perf_counter() - t1)
        # the same with the upload double-buffered on a copy stream (what SampleBlock does per window): batch n+1
        # travels while batch n is processed
        copy_stream = torch.cuda.Stream()
        bufs = [iq_d, torch.empty_like(iq_d)]
        used_by = [None, None]          # ticket of the step that last read the buffer

        def upload(i):
            if used_by[i] is not None:
                pipe.samples_consumed(used_by[i], stream=copy_stream)
            with torch.cuda.stream(copy_stream):
                bufs[i].copy_(iq_pin, non_blocking=True)

        fence()
        t2 = time.perf_counter()
        upload(0)
        for n in range(steps):
            cur = n & 1
            step(bufs[cur], in_stream=copy_stream)       # the lane waits for the upload enqueued so far on the copy stream
            used_by[cur] = last[0][0]
            if n + 1 < steps:
                upload(cur ^ 1)
        fence()
        pcie_overlapped = float(steps) * W * 2.0 * G * K / (time.perf_counter() - t2)

    out = None
    if rank == 0:
        units_per_step = W * 2.0 * G_global * K                    # (gridpoint, SV) pairs, both manifolds, all ranks
        value = units_per_step / (ms_step * 1e-3)
        windows_per_s = W / (ms_step * 1e-3)
        # --- roofline of the time-dominant kernel, SURVEY 8(d) algorithmic bytes per launch ON THIS RANK
        stage1 = bcs.stage1_kernel
        kname = {"bcm_scan": "bcm_scan_kernel", "bcs_bank": stage1, "bcs_finalize": "bcs_finalize_kernel",
                 "bcs_sum": "bcs_sum_kernel"}[dominant]
        alg = {"bcm_scan": 2 * W * 20.0 * G,            # 16 B grid read + 4 B score write per point, both manifolds in one launch
               "bcs_bank": Wl * 4.0 * S,                # every int16 I/Q sample once (all SVs of a window share the read)
               "bcs_sum": Wl * 4.0 * S,
               "bcs_finalize": Wl * K * (nLag + nBin) * 8.0}[dominant]
        ms_k, n_k = kern[dominant]
        avg_ms = ms_k / n_k if n_k else None
        ach = alg / (avg_ms * 1e-3) / 1e9 if avg_ms else 0.0
        whole_bytes = W * (4.0 * S + 2 * 20.0 * G_global)         # per step, samples counted once for the node
        traffic, tsrc = pmc_traffic(name, W, kname) if world == 1 else (None, None)
        out = {
            "metric": "manifold gridpoints x SVs correlated/sec", "value": value, "unit": "gridpoint*SV/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "headline": bool(headline),
            "config": {"workload": cfg["name"], "samples_per_window": S, "svs": K, "grid_points_per_manifold_per_gpu": G,
                       "grid_points_per_manifold_global": G_global, "manifolds": 2, "windows_per_step": W, "distinct_windows": distinct,
                       "lag_half_width": L, "bin_half_width": B, "exchange": args.exchange if ctx.use_dist else "none",
                       "comm": ("dpe_comm (C-ABI)" if ctx.comm is not None else "torch.distributed") if ctx.use_dist else "none",
                       "stage1": ("sharded by window + bank all-gather" if shard1 else "replicated") if ctx.use_dist else "local",
                       "scores_written": write_scores,
                       # batches in flight (dpe_pipe): lanes the steps of the timed region are dealt to; ms_per_step / value are
                       # THIS form's, one_stream_ms_per_step the same steps on one lane, the roofline stanza the isolated kernel
                       "in_flight": n_lanes,
                       # SURVEY 8(d) asks for batches of >= 256 resident windows; H is timed at 128 (2 MB each, 8 distinct ones repeated):
                       # per-window time measured equal at 64 / 128 / 256 windows per step (profiles/archive/r3a_H_w*_bench.json)
                       **({"windows_per_step_note": "128 per step against SURVEY 8(d)'s >= 256: same time per window at 64 / 128 / 256 "
                                                    "(profiles/archive/r3a_H_w64|w128|w256_bench.json)"} if name == "H" and W == 128 else {})},
            "one_stream_ms_per_step": one_stream_ms,
            "one_stream_value": units_per_step / (one_stream_ms * 1e-3),
            "timing": {"timed_batches": args.batches, "steps_per_timed_batch": reps * steps, "batch_ms_per_step": batch_ms,
                       "one_stream_batch_ms_per_step": one_ms,
                       "statistic": "median of the batches, each the max over ranks"},
            "x_realtime": windows_per_s / 50.0, "windows_per_s": windows_per_s,
            "roofline": {"bound": "hbm", "bound_physical": BOUND_PHYSICAL.get(kname, "valu"), "kernel": kname,
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": tsrc,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms,
                         "whole_step_algorithmic_bytes": whole_bytes,
                         "whole_step_frac": whole_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "algorithmic bytes (SURVEY 8d) over the kernel's HIP-event time; bound_physical names the "
                                 "measured limiter (SQ counters under profiles/), the kernels are not HBM-bound"},
            "kernels_ms_per_step": kernels_ms, "stage1_kernel": stage1, "stage1_dev_status": stage1_status,
            # the decoded ML grid points (global indices, scores) of the last step's first windows: the same for every rank
            # count that scans the same global grid (config M), which is what the multi-rank tests compare
            "fixes": fixes,
        }
        if pcie_value is not None:
            out["pcie_inclusive_value"] = pcie_value
            out["pcie_inclusive_overlapped_value"] = pcie_overlapped
        if world == 1 and headline:
            # measured ceiling beside the nominal peak: stream copy and triad over 1 GiB arrays
            cp, tr = dpe.engine.hbm_ceiling(1 << 30, 10, stream)
            out["roofline"]["measured_ceiling"] = {
                "copy_GBps": cp, "triad_GBps": tr, "algorithmic_over_triad": ach / tr,
                # what the kernel really pulls from HBM (PMC bytes / launch time): the grids are shared by the windows
                # of a batch and stay in L2, so the algorithmic rate may exceed the physical ceiling
                "physical_GBps": traffic / (avg_ms * 1e-3) / 1e9 if traffic and avg_ms else None}
    pipe.close()
    del iq_d, lanes, by_handle
    torch.cuda.empty_cache()
    return out, {"key": (fs, S, K, W), "data": (iq, cs, ce, bw)}


def brief(out):
    """The few numbers of a non-headline line that travel inside the headline JSON's `others` stanza (the driver records the last
    line only): time, value, the dominant kernel's roofline fractions, the device status and the CPU baseline of that configuration."""
    if out is None:
        return None
    r = out.get("roofline", {})
    b = {"ms_per_step": out["ms_per_step"], "value": out["value"], "unit": out["unit"], "x_realtime": out.get("x_realtime"),
         "roofline": {k: r.get(k) for k in ("kernel", "frac", "whole_step_frac", "avg_launch_ms", "flop_frac") if r.get(k) is not None}}
    if "modes" in out:      # acquisition: every mode's time and its fraction of the fp32 vector peak
        b["mode"] = out["config"].get("mode")
        b["modes"] = {m: {"ms": d["ms_per_window"], "flop_frac": d.get("flop_frac")} for m, d in out["modes"].items()}
        b["roofline"]["kernel"] = "dpe_acq_search"
    else:
        b["stage1_dev_status"] = out.get("stage1_dev_status")
        b["windows_per_step"] = out["config"]["windows_per_step"]
        for k in ("in_flight", "one_stream_ms_per_step"):
            if k in out.get("config", {}) or k in out:
                b[k] = out.get(k, out["config"].get(k))
    c = out.get("cpu_baseline")
    if c:
        b["cpu_baseline"] = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
    return b


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD (never exec: this process
    may not replace itself once anything touched the GPU, and it keeps the relay simple), pass its output through."""
    port = os.environ.get("MASTER_PORT", "29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--windows", type=int, default=None, help="windows per step (batch resident in HBM); default 256 (R, M) / 128 (H)")
    ap.add_argument("--config", choices=["R", "H", "M"], default=None,
                    help="time only this configuration.  R = BASELINE.json configs[1] (the metric's configuration, the default "
                         "headline); H = configs[2] (25 Msps, 12 SVs, 1e5-point grids); M = configs[3] (1e6-point global grids, "
                         "strong scaling over the GPUs).  Without it: H (N = 1) and M lines first, then the R headline")
    ap.add_argument("--no-extras", action="store_true", help="only the headline configuration")
    ap.add_argument("--extra-windows", type=int, default=None, help="windows per step of the extra (non-headline) lines")
    ap.add_argument("--comm", choices=["torch", "dpe"], default="torch",
                    help="who carries the N > 1 data-path exchange: torch.distributed (RCCL through PyTorch), or the C-ABI's own "
                         "dpe_comm (RCCL bound by libdpe_hip.so; host files when the ranks share one GPU in the functional tests) "
                         "-- dpe_bcs_allgather_banks + dpe_bcm_exchange_keys, what a C++ host (dpe_flow --ranks) uses")
    ap.add_argument("--exchange", choices=["keys", "scores"], default="keys",
                    help="multi-GPU exchange: packed arg-max keys (8 B/window/manifold) or the north-star-literal "
                         "all-reduce(SUM) of the zero-initialised full score vectors")
    ap.add_argument("--stage1", choices=["sharded", "replicated"], default="sharded",
                    help="N > 1: correlate W/N windows per rank and all-gather the banks (default), or recompute stage 1 everywhere")
    ap.add_argument("--acq", choices=["coherent", "noncoherent", "textbook"], default=None,
                    help="time the cold-start acquisition search instead (8f row 4; separate JSON line, not the headline)")
    ap.add_argument("--clock-warmup-s", type=float, default=1.0,
                    help="seconds of untimed steps after --warmup, before the timed region (GPU clock settle)")
    ap.add_argument("--batches", type=int, default=5, help="timed batches (the median is reported)")
    ap.add_argument("--min-batch-s", type=float, default=0.2, help="each timed batch repeats the --steps steps until it lasts this long")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=10.0,
                    help="seconds of one-thread oracle time behind the headline's cpu_baseline (H / M: 0.4 / 0.5 of it, at least one "
                         "whole window each); below 5 the all-cores leg is skipped")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight in the timed region (lanes of the library's dpe_pipe: stage 1 of step n + 1 beside the scan "
                         "of step n); 1 = one stream.  `one_stream_ms_per_step` is reported beside it either way")
    ap.add_argument("--no-pipelined", action="store_true", help="same as --in-flight 1 (kept for the profile collection scripts)")
    ap.add_argument("--no-scores", action="store_true", help="skip the per-point score write (arg-max only)")
    ap.add_argument("--include-h2d", action="store_true",
                    help="also time the steps with the window batch uploaded from pinned host memory inside the timed "
                         "region (reported as pcie_inclusive_value; never the headline value)")
    args = ap.parse_args()
    if args.no_pipelined:
        args.in_flight = 1
    if args.acq:
        return acq_main(args.acq)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args, sys.argv[1:])          # does not return

    import torch
    import torch.distributed as dist
    import navlab_dpe_sdr_amd as dpe  # noqa: F401

    ctx = Ctx()
    if ctx.world != args.gpus and not (ctx.world == 1 and args.gpus == 1):
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d ranks\n" % (args.gpus, ctx.world))
        sys.exit(2)
    if ctx.use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # DPE_BENCH_BACKEND=gloo DPE_BENCH_SHARE_GPU=1: functional check of the N > 1 code path on a one-GPU box (all
        # ranks on cuda:0, exchange through gloo) -- RCCL refuses two ranks on one device.  Never a performance number.
        if os.environ.get("DPE_BENCH_SHARE_GPU") == "1":
            ctx.local_rank = 0
        torch.cuda.set_device(ctx.local_rank)
        if ctx.backend == "nccl":
            dist.init_process_group("nccl", rank=ctx.rank, world_size=ctx.world, device_id=torch.device("cuda", ctx.local_rank))
        else:
            dist.init_process_group(ctx.backend, rank=ctx.rank, world_size=ctx.world)
        ctx.dist = dist
        if args.comm == "dpe":
            if args.exchange != "keys":
                sys.stderr.write("bench.py: --comm dpe carries the key exchange (--exchange keys) only\n")
                sys.exit(2)
            # rendezvous directory: the same for every rank of this launch, fresh per launch (rank 0's pid travels over the
            # control plane); dpe_comm itself trusts nothing it finds there by name
            tag = [os.getpid() if ctx.rank == 0 else 0]
            if ctx.world > 1:
                dist.broadcast_object_list(tag, src=0)
            rdv = os.path.join(os.environ.get("DPE_BENCH_RENDEZVOUS", "/tmp"), "dpe_bench_rdv_%s_%d" % (os.environ.get("MASTER_PORT", "0"), tag[0]))
            os.makedirs(rdv, exist_ok=True)
            shared_gpu = os.environ.get("DPE_BENCH_SHARE_GPU") == "1"
            ctx.comm = dpe.engine.Comm(ctx.rank, ctx.world, rdv, dpe.engine.Comm.HOSTFILES if shared_gpu else dpe.engine.Comm.RCCL)
    else:
        torch.cuda.set_device(0)
    ctx.dev = torch.device("cuda", ctx.local_rank if ctx.use_dist else 0)

    def emit(d):
        # RCCL writes a version banner through C stdio on stdout; push it out first so that the JSON line comes last
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if ctx.rank == 0 and d is not None:
            print(json.dumps(d), flush=True)

    headline = args.config or "R"
    cached = None
    configs = {"R": dpe.workload.CONFIG_R, "H": dpe.workload.CONFIG_H, "M": dpe.workload.CONFIG_M}
    want_cpu = ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline
    others = {}
    if args.config is None and not args.no_extras:
        extra_steps = min(args.steps, 20)
        if ctx.world == 1 and not ctx.use_dist:
            try:
                out = acq_line()                  # BASELINE.json configs[4]
                emit(out)
                others["acq"] = brief(out)
            except Exception as e:                # (an extra line must never cost the headline)
                sys.stderr.write("bench.py: acquisition line failed: %r\n" % (e,))
            out, hc = run_workload("H", ctx, args, extra_steps, min(args.warmup, 2), headline=False)
            if want_cpu:      # the CPU restatement beside every configuration; one thread, one whole window at H
                out["cpu_baseline"] = cpu_baseline(configs["H"], budget_s=0.4 * args.cpu_budget_s, all_cores=False, windows=hc["data"])
            emit(out)
            others["H"] = brief(out)
            del hc
        out, cached = run_workload("M", ctx, args, extra_steps, min(args.warmup, 2), headline=False)
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(configs["M"], budget_s=0.5 * args.cpu_budget_s, all_cores=False, windows=cached["data"])
        emit(out)
        if ctx.rank == 0:
            others["M"] = brief(out)
    out, _ = run_workload(headline, ctx, args, args.steps, args.warmup, headline=True, cached=cached)
    if want_cpu:
        cfg = configs[headline]
        out["cpu_baseline"] = cpu_baseline(dict(cfg, G=min(cfg["G"], 390625)), budget_s=args.cpu_budget_s,
                                           all_cores=args.cpu_budget_s >= 5.0)
    if ctx.rank == 0 and others:
        out["others"] = others        # the non-headline lines in brief: the driver records only this last line
    if ctx.use_dist:
        dist.barrier()
    emit(out)
    if ctx.comm is not None:
        ctx.comm.close()
    if ctx.use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
