"""The N > 1 path of bench.py (grid sharded over ranks, stage 1 sharded by window with a bank all-gather, per-step arg-max
exchange, max-over-ranks timing, JSON lines from rank 0 only) exercised with two ranks on ONE GPU: gloo carries the exchange
and both ranks use cuda:0 (RCCL refuses two ranks on one device).  A functional check of the code the driver launches with
--gpus 2/4/8 -- not a performance number.  Plus the 1-rank RCCL self-test: RCCL itself initialises and reduces on the box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--steps", "2", "--warmup", "1", "--clock-warmup-s", "0", "--min-batch-s", "0.01", "--batches", "2", "--no-cpu-baseline"]


def _launch(port, extra, nproc=2, expect_rc=0, timeout=300):
    env = dict(os.environ, DPE_BENCH_BACKEND="gloo", DPE_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + FAST + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    if expect_rc:
        assert r.returncode != 0
        return r.stderr
    assert r.returncode == 0, r.stderr[-3000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_eight_ranks_default_invocation_matches_one_rank():
    """What the driver launches on an 8-GPU node -- the default invocation with --gpus 8 (M strong scaling, then the R headline,
    weak scaling) -- with all eight ranks on this box's one GPU and gloo carrying the exchange: 8-way partition arithmetic of both
    grids, stage 1 sharded over 8 ranks (one window each), key all-reduce.  Config M scans the same global grid whatever the rank
    count, so its decoded fixes must be those of the one-rank run."""
    lines = _launch(29571, ["--windows", "8", "--extra-windows", "8"], nproc=8, timeout=900)
    assert len(lines) == 2, lines
    m, d = lines
    assert m["n_gpus"] == 8 and m["scaling"] == "strong" and m["config"]["stage1"].startswith("sharded")
    assert m["config"]["grid_points_per_manifold_global"] == 1000000 and m["config"]["grid_points_per_manifold_per_gpu"] == 125000
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["stage1"].startswith("sharded")
    assert d["config"]["grid_points_per_manifold_global"] == 8 * 390625 and d["config"]["grid_points_per_manifold_per_gpu"] == 390625
    assert d["config"]["windows_per_step"] == 8 and d["value"] > 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DPE_BENCH_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "M", "--windows", "8"] + FAST
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    one = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert one["n_gpus"] == 1 and one["config"]["grid_points_per_manifold_per_gpu"] == 1000000
    # rank 0 of 8 decodes the 1-rank run's ML points.  (Stage 1 runs with another tile partition when a rank holds one window
    # instead of eight: banks agree to fp32 rounding, so a window whose two best grid points tie within that rounding may pick
    # the other one -- its score then still agrees.)
    same = 0
    for a, b in zip(m["fixes"], one["fixes"]):
        assert abs(a[2] - b[2]) <= 2e-6 * abs(b[2]) and abs(a[3] - b[3]) <= 2e-6 * abs(b[3])
        same += (a[0] == b[0]) + (a[1] == b[1])
    assert len(m["fixes"]) == 8 and same >= 14


def test_windows_must_divide_over_the_ranks():
    """--stage1 sharded with a window count the ranks do not divide: exit status 2 and a message, not a silent fallback."""
    err = _launch(29572, ["--config", "R", "--windows", "3"], nproc=2, expect_rc=2)
    assert "divide over the 2 ranks" in err


@pytest.mark.parametrize("exchange,port", [("keys", 29561), ("scores", 29562)])
def test_two_rank_bench_lines(exchange, port):
    """Default invocation: the M line (strong scaling, 1e6-point global grids) then the R headline, rank 0 only."""
    lines = _launch(port, ["--windows", "4", "--extra-windows", "4", "--exchange", exchange])
    assert len(lines) == 2, lines
    m, d = lines
    assert m["headline"] is False and m["scaling"] == "strong" and m["n_gpus"] == 2 and m["value"] > 0
    assert m["config"]["grid_points_per_manifold_global"] == 1000000 and m["config"]["grid_points_per_manifold_per_gpu"] == 500000
    assert m["config"]["stage1"].startswith("sharded")
    assert d["headline"] is True and d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["exchange"] == exchange and d["config"]["windows_per_step"] == 4
    assert d["config"]["grid_points_per_manifold_global"] == 2 * 390625
    # (the stanza names the kernel that took the most time in the side pass: with four windows per step and two ranks sharing the
    # GPU the scan and the latency-bound small kernels are all ~10 us, so which one it is varies from run to run)
    assert d["roofline"]["kernel"] in ("bcm_scan_kernel", "bcs_finalize_kernel", "bcs_bank16_kernel", "bcs_bank_kernel", "bcs_sum_kernel")
    assert d["roofline"]["achieved"] > 0
    assert d["timing"]["timed_batches"] == 2 and len(d["timing"]["batch_ms_per_step"]) == 2


def test_two_rank_replicated_stage1_and_config_m_alone():
    lines = _launch(29563, ["--config", "M", "--windows", "4", "--stage1", "replicated"])
    assert len(lines) == 1 and lines[0]["headline"] is True and lines[0]["config"]["stage1"] == "replicated"
    assert lines[0]["scaling"] == "strong" and lines[0]["config"]["grid_points_per_manifold_per_gpu"] == 500000


def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher environment: bench.py starts torch.distributed.run as a child."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(DPE_BENCH_BACKEND="gloo", DPE_BENCH_SHARE_GPU="1", MASTER_PORT="29564")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "R", "--windows", "4"] + FAST
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_gpus_flag_must_match_the_launcher():
    env = dict(os.environ, DPE_BENCH_BACKEND="gloo", DPE_BENCH_SHARE_GPU="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "R", "--windows", "4"] + FAST
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2" in r.stderr


def test_one_rank_rccl_self_test():
    """DPE_BENCH_FORCE_DIST=1: one rank, backend nccl (= RCCL) -- the same exchange code path as N > 1; bench.py asserts that
    the all-reduced keys decode to the fix the handle itself reports."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DPE_BENCH_BACKEND")}
    env.update(DPE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29565", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "R", "--windows", "4"] + FAST
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["exchange"] == "keys" and d["value"] > 0
    assert d["config"]["stage1"].startswith("sharded")     # the banks went through all_gather_into_tensor on device views


def test_two_ranks_with_the_exchange_behind_the_c_abi():
    """--comm dpe: the bank all-gather and the key all-reduce of the timed step go through dpe_bcs_allgather_banks /
    dpe_bcm_exchange_keys (dpe_comm; host files here, where both ranks share the one GPU) instead of torch.distributed -- the
    exchange a C++ host (dpe_flow --ranks) uses.  Config M (the same global grid whatever the rank count): the decoded ML points
    equal the torch.distributed run's."""
    a = _launch(29566, ["--config", "M", "--windows", "4", "--comm", "dpe"])
    b = _launch(29567, ["--config", "M", "--windows", "4"])
    assert len(a) == 1 and len(b) == 1
    assert a[0]["config"]["comm"] == "dpe_comm (C-ABI)" and b[0]["config"]["comm"] == "torch.distributed"
    assert a[0]["config"]["stage1"].startswith("sharded") and a[0]["n_gpus"] == 2 and a[0]["value"] > 0
    assert a[0]["fixes"] == b[0]["fixes"]


def test_one_rank_rccl_self_test_through_the_c_abi():
    """DPE_BENCH_FORCE_DIST=1 --comm dpe: one rank, dpe_comm binds RCCL itself (ncclCommInitRank), all-gathers the banks and
    reduces the keys on the box; bench.py asserts that the exchanged keys decode to the handle's own fix."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DPE_BENCH_BACKEND")}
    env.update(DPE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29568", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "R", "--windows", "4", "--comm", "dpe"] + FAST
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["comm"] == "dpe_comm (C-ABI)" and d["value"] > 0


def test_default_invocation_carries_every_configuration_in_the_last_line():
    """`python bench.py` (N = 1): acq, H and M lines first, then the R headline whose `others` stanza repeats them in brief --
    time, value, roofline fractions, device status, CPU baseline -- because the driver records the last line only.  The timed
    regions run with two batches in flight (the library's dpe_pipe), the one-stream time beside them."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DPE_BENCH_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--clock-warmup-s", "0", "--min-batch-s", "0.01",
           "--batches", "2", "--cpu-budget-s", "0.5", "--windows", "32", "--extra-windows", "32"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 4 and [l["headline"] for l in lines] == [False, False, False, True]
    d = lines[-1]
    assert d["config"]["workload"].startswith("R:") and d["config"]["in_flight"] == 2 and d["one_stream_ms_per_step"] > 0
    assert d["roofline"]["kernel"] == "bcm_scan_kernel" and 0 < d["roofline"]["frac"] < 1.0
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    o = d["others"]
    assert set(o) == {"acq", "H", "M"}
    for k in ("H", "M"):
        assert o[k]["ms_per_step"] > 0 and o[k]["value"] > 0 and o[k]["in_flight"] == 2 and o[k]["one_stream_ms_per_step"] > 0
        assert o[k]["stage1_dev_status"] in (0, None) and 0 < o[k]["roofline"]["frac"] < 1.0 and o[k]["roofline"]["whole_step_frac"] > 0
        assert o[k]["cpu_baseline"]["kind"] == "port" and o[k]["cpu_baseline"]["cores"] == 1 and o[k]["cpu_baseline"]["value"] > 0
    assert o["H"]["roofline"]["kernel"] == "bcs_bank_chip2_kernel" and o["M"]["roofline"]["kernel"] == "bcm_scan_kernel"
    assert set(o["acq"]["modes"]) == {"coherent", "textbook", "noncoherent", "noncoherent_25x500Hz"}
    assert all(0 < m["flop_frac"] < 1 and m["ms"] > 0 for m in o["acq"]["modes"].values())
    assert o["acq"]["cpu_baseline"]["value"] > 0 and 0 < o["acq"]["roofline"]["flop_frac"] < 1
    # the brief copies are the lines' own numbers
    assert o["H"]["ms_per_step"] == lines[1]["ms_per_step"] and o["M"]["value"] == lines[2]["value"] and o["acq"]["value"] == lines[0]["value"]
