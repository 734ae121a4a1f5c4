"""The multi-GPU exchange behind the C-ABI (csrc/dpe_comm.hip, dpe_bcm_exchange_keys) and `dpe_flow --ranks`: the C++ flow
shards the manifold grid without Python.  Two ranks on ONE GPU go through the host-file transport (RCCL refuses two ranks
per device); RCCL itself is exercised with one rank: librccl bound at run time, communicator from a unique id, an
all-reduce(MAX) on device memory."""
import os
import subprocess

import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe

pytestmark = pytest.mark.gpu


def _inputs(tmp_path, W=4):
    fs, S, K = 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=7, amp=200.0)
    dat = str(tmp_path / "synthetic_2500kHz.dat")
    iq.tofile(dat)
    ho_path = str(tmp_path / "handoff.csv")
    with open(dpe.workload.HANDOFF_CSV) as f, open(ho_path, "w") as g:
        for line in f:
            g.write("bytes_read,0\n" if line.startswith("bytes_read") else line)
    return dat, ho_path


def test_two_flow_ranks_on_one_gpu_equal_the_unsharded_flow(tmp_path):
    W = 4
    dat, ho = _inputs(tmp_path, W)
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    base = [exe, "--samples", dat, "--handoff", ho, "--iters", str(W), "--grid-dim", "9", "--spacing", "1.0"]
    full = str(tmp_path / "X_full.csv")
    subprocess.check_call(base + ["--out", full], timeout=200)
    rdv = str(tmp_path / "rdv")
    os.makedirs(rdv)
    ref = np.loadtxt(full, delimiter=",")
    assert ref.shape == (W, 8)
    # twice through the SAME rendezvous directory: the second run finds the first one's exchange files (same names, same
    # sizes, stamped with another run's nonce) and must neither believe nor trip over them
    for attempt in range(2):
        outs = [str(tmp_path / ("X_run%d_rank%d.csv" % (attempt, r))) for r in range(2)]
        procs = [subprocess.Popen(base + ["--out", outs[r], "--ranks", "2", "--rank", str(r), "--rendezvous", rdv, "--comm", "files"],
                                  stderr=subprocess.PIPE, text=True) for r in range(2)]
        for p in procs:
            _, err = p.communicate(timeout=200)
            assert p.returncode == 0, err[-2000:]
        for o in outs:
            assert np.array_equal(np.loadtxt(o, delimiter=","), ref)       # every rank decodes the same global ML point


def test_two_flow_ranks_with_stage1_sharded_by_channel(tmp_path):
    """--shard-stage1: each flow correlates 4 of the 8 channels and dpe_bcs_allgather_banks (host-file transport here)
    completes the banks both grid shards are scored against; the fixes are the unsharded flow's."""
    W = 3
    dat, ho = _inputs(tmp_path, W)
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    base = [exe, "--samples", dat, "--handoff", ho, "--iters", str(W), "--grid-dim", "9", "--spacing", "1.0"]
    full = str(tmp_path / "X_full.csv")
    subprocess.check_call(base + ["--out", full], timeout=200)
    rdv = str(tmp_path / "rdv")
    os.makedirs(rdv)
    outs = [str(tmp_path / ("X_s1_rank%d.csv" % r)) for r in range(2)]
    procs = [subprocess.Popen(base + ["--out", outs[r], "--ranks", "2", "--rank", str(r), "--rendezvous", rdv, "--comm", "files",
                                      "--shard-stage1"], stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        _, err = p.communicate(timeout=200)
        assert p.returncode == 0, err[-2000:]
    ref = np.loadtxt(full, delimiter=",")
    for o in outs:
        assert np.array_equal(np.loadtxt(o, delimiter=","), ref)
    assert os.path.isdir(os.path.join(rdv, "stage1"))        # the stage-1 exchange really had its own rendezvous


def test_flow_with_one_rccl_rank(tmp_path):
    """--ranks 1 --comm rccl: ncclCommInitRank + ncclAllReduce(MAX, uint64) through the C-ABI on the box's GPU."""
    W = 3
    dat, ho = _inputs(tmp_path, W)
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    base = [exe, "--samples", dat, "--handoff", ho, "--iters", str(W), "--grid-dim", "9", "--spacing", "1.0"]
    full, one = str(tmp_path / "X_full.csv"), str(tmp_path / "X_rccl.csv")
    subprocess.check_call(base + ["--out", full], timeout=200)
    rdv = str(tmp_path / "rdv")
    os.makedirs(rdv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(base + ["--out", one, "--ranks", "1", "--rank", "0", "--rendezvous", rdv, "--comm", "rccl"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.loadtxt(one, delimiter=","), np.loadtxt(full, delimiter=","))


def test_sharded_device_loop_equals_the_unsharded_device_loop(tmp_path):
    """dpe_flow --device-loop --ranks N: the device-resident channel manager with the grid sharded (dpe_chm_dev_set_shard) -- per window
    update_prepared -> update_prepared -> all-reduce(MAX) of the device keys -> measurement kernel on the reduced keys against the
    global grids (the reference takes its arg-max at batchcorrmanifold.cu:2589-2596).  Two ranks on one GPU through the host-file
    transport and one rank through RCCL (stream-ordered ncclAllReduce on the device keys, nothing read back) write the X-file rows
    of the unsharded device loop, and every rank ends with the same fix.  No scaling number is claimed."""
    W = 12
    dat, ho = _inputs(tmp_path, W)
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    base = [exe, "--samples", dat, "--handoff", ho, "--iters", str(W), "--grid-dim", "9", "--spacing", "1.0", "--init-delta", "2", "-1", "1", "3",
            "--device-loop", "--fix-lag", "3"]
    full = str(tmp_path / "X_dev.csv")
    subprocess.check_call(base + ["--out", full], timeout=200)
    ref = np.loadtxt(full, delimiter=",")
    assert ref.shape == (W, 8)
    rdv = str(tmp_path / "rdv")
    os.makedirs(rdv)
    outs = [str(tmp_path / ("X_dev_rank%d.csv" % r)) for r in range(2)]
    procs = [subprocess.Popen(base + ["--out", outs[r], "--ranks", "2", "--rank", str(r), "--rendezvous", rdv, "--comm", "files"],
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        _, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-2000:]
    for o in outs:
        assert np.array_equal(np.loadtxt(o, delimiter=","), ref)
    rdv1 = str(tmp_path / "rdv1")
    os.makedirs(rdv1)
    one = str(tmp_path / "X_dev_rccl.csv")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(base + ["--out", one, "--ranks", "1", "--rank", "0", "--rendezvous", rdv1, "--comm", "rccl"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.loadtxt(one, delimiter=","), ref)


def test_comm_allreduce_max_through_python(tmp_path):
    """dpe_comm_allreduce_max_u64 on a device buffer: one RCCL rank (identity), and the packed-key semantics of the
    host-file transport with two communicator objects driven from two threads."""
    import threading
    import torch
    c = dpe.engine.Comm(0, 1, str(tmp_path), dpe.engine.Comm.RCCL)
    t = torch.arange(16, dtype=torch.int64, device="cuda:0") * 1000003
    c.allreduce_max_u64(t.data_ptr(), 16)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(16, dtype=torch.int64) * 1000003)
    c.close()
    a = torch.tensor([5, 1, 9, 2], dtype=torch.int64, device="cuda:0")
    b = torch.tensor([3, 7, 9, 8], dtype=torch.int64, device="cuda:0")
    comms = [None, None]     # the rendezvous is a handshake: both ranks have to be in it at the same time

    def make(r):
        comms[r] = dpe.engine.Comm(r, 2, str(tmp_path), dpe.engine.Comm.HOSTFILES)

    th = [threading.Thread(target=make, args=(r,)) for r in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join(60)
    assert all(comms)
    th = [threading.Thread(target=comms[r].allreduce_max_u64, args=(x.data_ptr(), 4)) for r, x in enumerate((a, b))]
    for x in th:
        x.start()
    for x in th:
        x.join(60)
    assert a.cpu().tolist() == [5, 7, 9, 8] and b.cpu().tolist() == [5, 7, 9, 8]
    for x in comms:
        x.close()
