"""The N > 1 path of bench.py (grid sharded over ranks, per-step exchange, max-over-ranks timing, one JSON line from rank 0)
exercised with two ranks on ONE GPU: gloo carries the exchange and both ranks use cuda:0 (RCCL refuses two ranks on one
device).  A functional check of the code the driver launches with --gpus 2/4/8 -- not a performance number."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("exchange,port", [("keys", 29561), ("scores", 29562)])
def test_two_rank_bench_line(exchange, port):
    env = dict(os.environ, DPE_BENCH_BACKEND="gloo", DPE_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--clock-warmup-s", "0", "--windows", "4", "--exchange", exchange, "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["exchange"] == exchange and d["config"]["windows_per_step"] == 4
    assert d["roofline"]["kernel"] == "bcm_scan_kernel" and d["roofline"]["achieved"] > 0
