"""Host-side logic of bench.py and the sharded workloads that needs no GPU."""
import json
import os
import subprocess
import sys

import numpy as np

import navlab_dpe_sdr_amd as dpe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_flag_must_match_the_launcher_environment():
    """--gpus N with a launcher that started a different number of ranks: status 2 before anything touches a GPU."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 4" in r.stderr and not r.stdout.strip()


def test_strong_scaling_shards_partition_the_global_grid():
    G = 1000003
    pos_g, vel_g, p0, v0, off0 = dpe.workload.build_grids_strong(G, 0, 8)
    sizes, nxt = [], 0
    for r in range(8):
        _, _, p, v, off = dpe.workload.build_grids_strong(G, r, 8)
        assert off == nxt and p.shape == v.shape and np.array_equal(p, pos_g[off:off + p.shape[0]])
        nxt = off + p.shape[0]
        sizes.append(p.shape[0])
    assert nxt == G and max(sizes) - min(sizes) <= 1
    assert (0, G) == dpe.sharding.shard_range(G, 0, 1)


def test_weak_scaling_shards_are_slices_of_one_global_grid():
    a = [dpe.workload.build_grids(1000, r, 4) for r in range(4)]
    for r, (pg, vg, p, v, off) in enumerate(a):
        assert off == 1000 * r and np.array_equal(pg, a[0][0]) and np.array_equal(p, pg[off:off + 1000])


def test_committed_pmc_traffic_is_found_per_configuration():
    import bench
    for cfg, w, kern, lo, hi in (("R", 256, "bcm_scan_kernel", 0.5e9, 2e9),
                                 ("H", 128, "bcs_bank_chip2_kernel", 2.5e8, 7e8), ("M", 256, "bcm_scan_kernel", 1e9, 4e9)):
        t, src = bench.pmc_traffic(cfg, w, kern)
        assert t is not None and lo < t < hi and src.startswith("profiles/") and os.path.exists(os.path.join(ROOT, src))
        assert json.load(open(os.path.join(ROOT, src)))["config"] == cfg
    assert bench.pmc_traffic("R", 7, "bcm_scan_kernel") == (None, None)       # no profile for that batch size
    assert bench.pmc_traffic("H", 32, "bcs_bank_chip_kernel") == (None, None)  # round 2's shape: its files are under profiles/archive/, not looked up


def test_config_table():
    w = dpe.workload
    assert w.CONFIG_R["S"] == 50000 and w.CONFIG_R["G"] == 25 ** 4 and w.CONFIG_R["K"] == 8
    assert w.CONFIG_H["fs"] == 25e6 and w.CONFIG_H["S"] == 500000 and w.CONFIG_H["K"] == 12 and w.CONFIG_H["G"] == 100000
    assert w.CONFIG_M["G"] == 1000000 and w.CONFIG_M["fs"] == 2.5e6
    # the bank widths the bench uses cover the rngrid3-format grids
    for c in (w.CONFIG_R, w.CONFIG_H, w.CONFIG_M):
        pos = dpe.synth.rand_grid(3, 4096)
        vel = dpe.synth.rand_grid(4, 4096, half=(6.0, 6.0, 6.0, 3.0))
        L, B = dpe.pipeline.bank_half_widths(pos, vel, c["fs"], dpe.engine.carr_fft_len(c["S"]))
        assert L <= c["L"] + 2 and B <= c["B"] + 3


OTHERS_KEYS = {"ms_per_step", "value", "unit", "x_realtime", "roofline", "cpu_baseline"}


def test_others_stanza_schema():
    """The headline JSON line carries the non-headline configurations in brief (`others`: acq, H, M) -- the driver records the
    last line only.  bench.brief() on lines of the shape run_workload / acq_line produce."""
    import bench
    cpu = {"value": 1.0e7, "unit": "gridpoint*SV/s", "cores": 1, "kind": "port", "sample": "1 full windows", "x_realtime": 0.01, "host": {}}
    line = {"ms_per_step": 0.65, "value": 4.7e11, "unit": "gridpoint*SV/s", "x_realtime": 3900.0, "stage1_dev_status": 0,
            "one_stream_ms_per_step": 0.71,
            "config": {"workload": "H: ...", "windows_per_step": 128, "in_flight": 2},
            "roofline": {"bound": "hbm", "kernel": "bcs_bank_chip2_kernel", "frac": 0.058, "whole_step_frac": 0.14, "avg_launch_ms": 0.55,
                         "achieved": 465.0, "peak": 8000.0, "note": "long text that must not travel"},
            "cpu_baseline": cpu, "kernels_ms_per_step": {"bcs_bank": 0.55}, "fixes": [[1, 2, 3.0, 4.0]] * 16}
    b = bench.brief(line)
    assert OTHERS_KEYS | {"stage1_dev_status", "windows_per_step", "in_flight", "one_stream_ms_per_step"} <= set(b)
    assert set(b["roofline"]) == {"kernel", "frac", "whole_step_frac", "avg_launch_ms"}
    assert set(b["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and b["in_flight"] == 2
    acq = {"ms_per_step": 0.055, "value": 1.8e11, "unit": "cell/s", "x_realtime": 180.0, "config": {"workload": "acq: ...", "mode": "coherent"},
           "modes": {m: {"ms_per_window": t, "cells_per_s": 1.0, "flop_frac": f} for m, t, f in
                     (("coherent", 0.055, 0.07), ("textbook", 0.24, 0.15), ("noncoherent", 0.30, 0.15), ("noncoherent_25x500Hz", 0.11, 0.1))},
           "roofline": {"bound": "hbm", "kernel": "dpe_acq_search (a long description)", "frac": 0.09, "flop_frac": 0.075, "avg_launch_ms": 0.055},
           "cpu_baseline": {"value": 3.6e6, "unit": "cell/s", "cores": 1, "kind": "port", "sample": "x"}}
    a = bench.brief(acq)
    assert OTHERS_KEYS | {"modes", "mode"} <= set(a) and set(a["modes"]) == set(acq["modes"])
    assert all(set(v) == {"ms", "flop_frac"} for v in a["modes"].values()) and a["roofline"]["flop_frac"] == 0.075
    # short enough to survive beside the headline's own fields
    assert len(json.dumps({"acq": a, "H": b, "M": b})) < 2500
    assert bench.brief(None) is None
    assert bench.FP32_VECTOR_PEAK_TFLOPS == 157.3 and bench.HBM_PEAK_GBS == 8000.0
