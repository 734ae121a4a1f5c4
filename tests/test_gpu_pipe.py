"""Batches in flight (csrc/dpe_pipe.hip, include/dpe_hip.h "batches in flight"): a dpe_pipe deals consecutive batches to lanes on
their own streams.  Every lane runs the launches a lone handle pair would, so everything a batch produces -- banks, per-point
scores, arg-max keys, the measurement -- must be BIT-IDENTICAL to the one-stream path, also when batches with different contents
are interleaved without a host wait in between.  Shapes: the three configurations bench.py times (R, H, M)."""
import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe

pytestmark = pytest.mark.gpu


def _view(ptr, shape, typestr):
    import torch

    class _Cai:
        __cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Cai(), device="cuda:0")


def _same_results(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x["posIndex"] == y["posIndex"] and x["velIndex"] == y["velIndex"]
        assert x["posScore"] == y["posScore"] and x["velScore"] == y["velScore"]
        assert x["posOutOfWindow"] == y["posOutOfWindow"] and x["velOutOfWindow"] == y["velOutOfWindow"]
        assert np.array_equal(x["zVal"], y["zVal"])


def _workload(name, W, distinct):
    cfg = {"R": dpe.workload.CONFIG_R, "H": dpe.workload.CONFIG_H, "M": dpe.workload.CONFIG_M}[name]
    fs, S, K = cfg["fs"], cfg["S"], cfg["K"]
    iq, cs, ce, bw = dpe.workload.build_windows(distinct, fs, S, K, seed=11, amp=cfg["amp"])
    rep = (W + distinct - 1) // distinct
    iq, cs, ce, bw = (np.concatenate([a] * rep)[:W] for a in (iq, cs, ce, bw))
    if name == "M":
        pos, vel, _, _, _ = dpe.workload.build_grids_strong(cfg["G"], 0, 1)
    else:
        _, _, pos, vel, _ = dpe.workload.build_grids(cfg["G"])
    return cfg, (iq, cs, ce, bw), pos, vel


@pytest.mark.parametrize("name,W,distinct", [("R", 256, 16), ("H", 128, 4), ("M", 256, 16)])
def test_interleaved_batches_equal_the_one_stream_path(name, W, distinct):
    import torch
    cfg, (iq, cs, ce, bw), pos, vel = _workload(name, W, distinct)
    fs, S, K, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["L"], cfg["B"]
    # two batches with different contents: B is A rolled by one window
    batches = [(iq, cs, ce, bw), tuple(np.roll(a, 1, axis=0) for a in (iq, cs, ce, bw))]
    iq_d = [torch.from_numpy(np.ascontiguousarray(b[0])).to("cuda:0") for b in batches]
    nLag, nBin = 2 * L + 1, 2 * B + 1
    # ---- the one-stream path
    bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K)
    bcm.Start()
    ref = []
    for i, b in enumerate(batches):
        bcs.Update(iq_d[i], b[1])
        bcm.Update(bcs.CodeScores, bcs.CarrScores, b[3], b[2])
        res = bcm.results()
        ref.append(dict(res=res,
                        code=_view(bcs.CodeScores, (W, K, nLag, 2), "<f4").clone(), carr=_view(bcs.CarrScores, (W, K, nBin, 2), "<f4").clone(),
                        ps=_view(bcm.PosScores, (W, bcm.PosScoresPitch), "<f4")[:, :pos.shape[0]].clone(),
                        vs=_view(bcm.VelScores, (W, bcm.VelScoresPitch), "<f4")[:, :vel.shape[0]].clone(),
                        kernel=bcs.stage1_kernel))
    bcm.Stop(); bcs.Stop()
    assert ref[0]["res"][0]["posIndex"] != ref[1]["res"][0]["posIndex"] or not np.array_equal(ref[0]["res"][0]["zVal"], ref[1]["res"][0]["zVal"]) \
        or distinct == 1
    # ---- two batches in flight: A B A B A B enqueued back to back, each collected only when its lane is about to be dealt again
    pipe = dpe.Pipe(fs, S, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K, in_flight=2)
    st = torch.cuda.current_stream()
    tickets = []
    for n in range(6):
        i = n & 1
        if n >= 2:      # the lane of ticket n - 2 is dealt again by this submit: collect it first
            t_old, i_old = tickets[n - 2]
            _same_results(pipe.results(t_old), ref[i_old]["res"])
            fb, fm, _ = pipe.lane(t_old)
            assert fb.stage1_kernel == ref[i_old]["kernel"]
            assert torch.equal(_view(fb.CodeScores, (W, K, nLag, 2), "<f4"), ref[i_old]["code"])
            assert torch.equal(_view(fb.CarrScores, (W, K, nBin, 2), "<f4"), ref[i_old]["carr"])
            assert torch.equal(_view(fm.PosScores, (W, fm.PosScoresPitch), "<f4")[:, :pos.shape[0]], ref[i_old]["ps"])
            assert torch.equal(_view(fm.VelScores, (W, fm.VelScoresPitch), "<f4")[:, :vel.shape[0]], ref[i_old]["vs"])
        b = batches[i]
        tickets.append((pipe.submit(iq_d[i], b[1], b[3], b[2], stream=st), i))
    for t, i in tickets[-2:]:
        _same_results(pipe.results(t), ref[i]["res"])
    # a ticket whose lane has been dealt again is refused, not answered with another batch's results
    with pytest.raises(dpe.DpeError, match="is gone"):
        pipe.results(tickets[0][0])
    with pytest.raises(dpe.DpeError, match="never issued"):
        pipe.results(99)
    # ... but the ring's "slot free again" may be asked about it (a ring deeper than the lanes): it is ordered behind the later batch of
    # that lane; a ticket that was never issued is refused there too
    pipe.samples_consumed(tickets[0][0], stream=st)
    with pytest.raises(dpe.DpeError, match="never issued"):
        pipe.samples_consumed(99, stream=st)
    pipe.close()


def test_acquire_commit_form_and_stream_ordered_hand_backs():
    """The pieces a multi-GPU host uses (acquire -> its own Updates on the lane's stream -> commit) give the same results as
    submit; dpe_pipe_samples_consumed orders a refill of the sample buffer behind stage 1 (the SampleBlock ring's slot reuse,
    sampleblock.cu:421-447) and dpe_pipe_join orders a consumer behind everything in flight -- with no host wait anywhere."""
    import torch
    cfg, (iq, cs, ce, bw), pos, vel = _workload("R", 32, 8)
    fs, S, K, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["L"], cfg["B"]
    G = 20000
    pos, vel = pos[:G], vel[:G]
    W = 32
    pipe = dpe.Pipe(fs, S, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K, in_flight=2)
    assert dpe.engine.lib().dpe_pipe_in_flight(pipe._h) == 2
    st = torch.cuda.current_stream()
    buf = torch.from_numpy(iq).to("cuda:0")
    other = torch.from_numpy(np.ascontiguousarray(np.roll(iq, 3, axis=0))).to("cuda:0")
    t0 = pipe.submit(buf, cs, bw, ce, stream=st)
    want = pipe.results(t0)
    # refill behind stage 1: the copy into `buf` waits (on the device) for the lane to have read it
    t1 = pipe.submit(buf, cs, bw, ce, stream=st)
    pipe.samples_consumed(t1, stream=st)
    buf.copy_(other, non_blocking=True)
    _same_results(pipe.results(t1), want)
    # acquire / commit with the same Updates issued by the host
    buf.copy_(torch.from_numpy(iq).to("cuda:0"))
    t2, fb, fm, lane_stream = pipe.acquire(stream=st)
    fb.Update(buf, cs, stream=lane_stream)
    pipe.mark_stage1(t2)
    fm.Update(fb.CodeScores, fb.CarrScores, bw, ce, stream=lane_stream)
    with pytest.raises(dpe.DpeError, match="not committed"):
        pipe.results(t2)
    pipe.commit(t2, W, K)
    with pytest.raises(dpe.DpeError, match="committed before"):
        pipe.commit(t2, W, K)
    # join: a consumer on the caller's stream sees the finished keys without any host synchronisation
    pipe.join(stream=st)
    keys = _view(fm.Keys, (W, 2), "<i8").clone()          # enqueued on `st`, behind the join
    _same_results(pipe.results(t2), want)
    dec = fm.results_from_keys(keys.cpu().numpy().view(np.uint64), pos, vel)
    assert all(d["posIndex"] == w["posIndex"] and d["velIndex"] == w["velIndex"] for d, w in zip(dec, want))
    # an acquired lane that was never committed blocks the NEXT deal of that lane, loudly
    t3 = pipe.acquire(stream=st)[0]
    t4 = pipe.submit(buf, cs, bw, ce, stream=st)
    with pytest.raises(dpe.DpeError, match="never committed"):
        pipe.submit(buf, cs, bw, ce, stream=st)
    pipe.commit(t3, W, K)
    _same_results(pipe.results(t4), want)
    pipe.synchronize()
    pipe.close()


def test_pipe_argument_errors():
    cfg = dpe.workload.CONFIG_R
    pos = dpe.synth.rand_grid(3, 4096)
    vel = dpe.synth.rand_grid(4, 4096, half=(6.0, 6.0, 6.0, 3.0))
    with pytest.raises(dpe.DpeError, match="inFlight"):
        dpe.Pipe(cfg["fs"], cfg["S"], pos, vel, lag_half_width=4, bin_half_width=20, max_windows=2, max_channels=8, in_flight=0)
    with pytest.raises(dpe.DpeError, match="inFlight"):
        dpe.Pipe(cfg["fs"], cfg["S"], pos, vel, lag_half_width=4, bin_half_width=20, max_windows=2, max_channels=8, in_flight=9)
    p = dpe.Pipe(cfg["fs"], cfg["S"], pos, vel, lag_half_width=4, bin_half_width=20, max_windows=2, max_channels=8, in_flight=3)
    with pytest.raises(dpe.DpeError, match="never issued"):
        p.lane(0)
    p.close()


def test_cpp_host_drives_the_pipe(tmp_path):
    """host/test_pipe.cpp: a C++ program that sees nothing but include/dpe_hip.h replays six batches open-loop -- a three-slot ring of
    pinned blocks and device slots in front of two lanes, refills ordered by dpe_pipe_samples_consumed, each batch collected while the
    next one runs -- and compares every result with the one-stream calls (dpe_bcs_update + dpe_bcm_update) bit for bit."""
    import os
    import subprocess
    cfg = dpe.workload.CONFIG_R
    fs, S, K, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["L"], cfg["B"]
    W, nB, G = 8, 6, 10000
    iq, cs, ce, bw = dpe.workload.build_windows(W * nB, fs, S, K, seed=21, amp=cfg["amp"])
    _, _, pos, vel, _ = dpe.workload.build_grids(G)
    (tmp_path / "meta.txt").write_text("%r %d %d %d %d %d %d %d %d\n" % (fs, S, K, W, nB, L, B, pos.shape[0], vel.shape[0]))
    iq.tofile(tmp_path / "iq.bin")
    np.ascontiguousarray(cs).tofile(tmp_path / "cs.bin")
    np.ascontiguousarray(ce).tofile(tmp_path / "ce.bin")
    np.ascontiguousarray(bw).tofile(tmp_path / "win.bin")
    np.ascontiguousarray(pos, dtype=np.float64).tofile(tmp_path / "pos.bin")
    np.ascontiguousarray(vel, dtype=np.float64).tofile(tmp_path / "vel.bin")
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "test_pipe")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "6 batches x 8 windows, 0 results differ" in r.stdout
