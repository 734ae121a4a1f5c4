"""Device-resident channel manager (dpe_chm_dev_*, csrc/dpe_chanmgr.hip) -- the reference's own form of cuChanMgr: state and
port arrays in device memory, one small kernel per window (cudarecv/modules/src/cuchanmgr.cu:1100-1132,1237-1264) -- against
the host form dpe_chm_* (itself pinned by the oracle and fixtures O4 / O5 / O7, tests/test_abi_cpu.py), and the closed loop that
reads nothing back per window against the host-driven loop.

Tolerances.  The two forms share their functions; what differs is the maths library (sin / cos / atan2 of the device against
glibc's: last-bit differences) and the satellite states: the device form starts its Kepler iterations from the previous
window's anomaly (same fixed point) and takes a window's second state from the first along a difference quotient (remainder
3e-11 m).  Frequencies, phases, satellite states, ENU matrix: 1e-12 relative.  Code phases:
the reference forms them as (rxTime - pr / C - ...) x 1.023e6 with rxTime ~ 4e5 s, i.e. on a 5.8e-11 s = 6e-5 chip raster
(DESIGN.md 2.5); a last-bit difference in pr can move the result by one step of that raster, so they are held to 1e-4 chips
(in the cases below they agree to 1e-9)."""
import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe

pytestmark = pytest.mark.gpu


def _compare(dev, host, K):
    sd, ed, wd, bd = dev
    sh, eh, wh, bh = host
    for f in ("cpElapsedStart", "cpReference", "prn"):
        assert np.array_equal(sd[f], sh[f]), f
    for f in ("cpRefTOW", "cpElapsedEnd", "cpRef"):
        assert np.array_equal(ed[f], eh[f]), f
    worst = {}
    for name, a, b, tol, rel in (("rcStart", sd["codePhaseStart"], sh["codePhaseStart"], 1e-4, False),
                                 ("rcEnd", ed["codePhaseEnd"], eh["codePhaseEnd"], 1e-4, False),
                                 ("riStart", sd["carrierPhaseStart"], sh["carrierPhaseStart"], 1e-9, False),
                                 ("fc", sd["codeFrequency"], sh["codeFrequency"], 1e-12, True),
                                 ("fi", sd["carrierFrequency"], sh["carrierFrequency"], 1e-9, False),
                                 ("R", wd["enu2ecef"], wh["enu2ecef"], 1e-14, False),
                                 ("x", wd["xCurrkk1"], wh["xCurrkk1"], 0.0, False)):
        err = np.abs(np.asarray(a) - np.asarray(b)).max()
        if rel:
            err /= np.abs(np.asarray(b)).max()
        worst[name] = err
        assert err <= tol, (name, err)
    assert wd["rxTime"][0] == wh["rxTime"][0] and wd["dopplerSign"][0] == wh["dopplerSign"][0]
    # batch satellite states: positions relative to the orbit radius, velocities to the speed, clock terms absolutely
    pos = np.abs(bd[..., :3] - bh[..., :3]).max() / 2.6e7
    vel = np.abs(bd[..., 4:7] - bh[..., 4:7]).max() / 3.0e3
    clk = max(np.abs(bd[..., 3] - bh[..., 3]).max(), np.abs(bd[..., 7] - bh[..., 7]).max())
    assert pos < 1e-12 and vel < 1e-12 and clk < 1e-17, (pos, vel, clk)
    assert np.array_equal(ed["satState"], bd[:, bd.shape[1] // 2])
    worst.update(satPos=pos, satVel=vel)
    return worst


@pytest.mark.parametrize("K", [8, 1])
def test_device_channel_manager_matches_the_host_form(K):
    """Start + 5 Updates with a moving fix (tens of metres and m/s per window, clock terms included), time grid of 9 entries:
    every port of the device form against dpe_chm_outputs."""
    import torch
    ho = dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV)
    T = 0.02
    tg = np.linspace(-12.0, 12.0, 9)
    host = dpe.engine.ChanMgr.from_handoff(ho, T, K)
    dev = dpe.engine.ChanMgrDev.from_handoff(ho, T, K, tg)
    x = np.array(ho["X_ECEF"], dtype=np.float64).copy()
    host.Start(x, x, tg)
    dev.Start(x)
    w = _compare(dev.outputs(with_batch=True), host.outputs(with_batch=True), K)
    rng = np.random.Generator(np.random.PCG64(5))
    for it in range(5):
        x1 = x + np.concatenate([rng.uniform(-30, 30, 3), rng.uniform(-40, 40, 1), rng.uniform(-3, 3, 3), rng.uniform(-1, 1, 1)])
        xk = x1 + np.concatenate([rng.uniform(-2, 2, 4), rng.uniform(-0.2, 0.2, 4)])
        host.Update(x1, xk, tg)
        a, b = torch.from_numpy(x1).to("cuda:0"), torch.from_numpy(xk).to("cuda:0")
        dev.Update(a, b)
        w = _compare(dev.outputs(with_batch=True), host.outputs(with_batch=True), K)
        assert dev.status == 0, dev.status     # no Kepler failure
        x = x1
    print("worst differences after 5 Updates:", {k: float("%.3g" % v) for k, v in w.items()})
    dev.Stop(); host.Stop()


def test_device_closed_loop_matches_the_host_driven_loop():
    """40 windows of a receiver moving at (5, -3, 2) m/s, 9^4-point grids, 8 SVs: the loop that reads nothing back
    (UpdatePrepared x 2 + ChanMgrDev.step per window, fixes from the pinned ring, the host up to 7 windows ahead) gives the
    grid points of the host-driven loop window by window, and fixes equal to 1e-8 m (the ENU matrix comes from the device's
    atan2 / sincos instead of glibc's: last-bit differences)."""
    fs, S, K, W = 2.5e6, 50000, 8, 40
    v = np.array([5.0, -3.0, 2.0])
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=71, amp=200.0, velocity=v)
    ho = dpe.workload.extend_handoff(dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV), K)
    pos = dpe.synth.uniform_grid(9, 1.0)
    vel = dpe.synth.uniform_grid(9, 2.0)
    tg = np.unique(pos[:, 3])
    fixes_h, res_h = dpe.pipeline.run_closed_loop(iq, ho, fs, pos, vel, time_grid=tg, K=K)
    fixes_d, res_d, status = dpe.pipeline.run_device_loop(iq, ho, fs, pos, vel, time_grid=tg, K=K, ring_depth=8)
    assert status == 0
    for w in range(W):
        assert res_d[w]["posIndex"] == res_h[w]["posIndex"] and res_d[w]["velIndex"] == res_h[w]["velIndex"], w
        assert res_d[w]["posOutOfWindow"] == res_h[w]["posOutOfWindow"] and res_d[w]["velOutOfWindow"] == res_h[w]["velOutOfWindow"]
        assert abs(res_d[w]["posScore"] - res_h[w]["posScore"]) <= 2e-6 * res_h[w]["posScore"]
    assert np.abs(fixes_d - fixes_h).max() < 1e-8
    verr = np.linalg.norm(fixes_d[:, 4:7] - v, axis=1)
    assert verr[0] > 5.0 and verr[15:].max() < 2.0       # the loop converges as the host-driven one does


def test_device_ports_feed_the_dev_update_forms():
    """A host that keeps the reference's module structure: ChanMgrDev's port arrays handed to dpe_bcs_update_dev /
    dpe_bcm_update_dev (prep kernels) instead of the prepared forms -- same result for a window."""
    import torch
    fs, S, K = 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(1, fs, S, K, seed=3, amp=200.0)
    ho = dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV)
    g = dpe.synth.uniform_grid(9, 1.0)
    tg = np.unique(g[:, 3])
    L, B = dpe.pipeline.bank_half_widths(g, g, fs, dpe.engine.carr_fft_len(S))
    x = np.array(ho["X_ECEF"], dtype=np.float64)
    iq_d = torch.from_numpy(iq).to("cuda:0")
    out = []
    for form in ("prepared", "ports"):
        bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_channels=K)
        bcs.Start()
        bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, g, g, lag_half_width=L, bin_half_width=B, max_channels=K)
        bcm.Start()
        cm = dpe.engine.ChanMgrDev.from_handoff(ho, S / fs, K, tg)
        if form == "prepared":
            cm.attach(bcs, bcm, 4)
            cm.Start(x)
            bcs.UpdatePrepared(iq_d[0], K)
            bcm.UpdatePrepared(bcs.CodeScores, bcs.CarrScores, K)
        else:
            cm.Start(x)
            pb, pm, rx, _, _, _ = cm.ports()
            s, e, win = cm.outputs()
            bcs.UpdateDev(iq_d[0], K, {n: getattr(pb, n) for n, _ in pb._fields_})
            bcm.UpdateDev(bcs.CodeScores, bcs.CarrScores, K, {n: getattr(pm, n) for n, _ in pm._fields_ if n not in ("dimT", "reserved")},
                          tg.size, float(win["rxTime"][0]))
        out.append(bcm.results()[0])
        cm.Stop(); bcm.Stop(); bcs.Stop()
    a, b = out
    assert a["posIndex"] == b["posIndex"] and a["velIndex"] == b["velIndex"] and a["posScore"] == b["posScore"] and a["velScore"] == b["velScore"]
    assert np.array_equal(a["zVal"], b["zVal"])


def test_cpp_flow_device_loop_matches_the_host_driven_flow(tmp_path):
    """host/dpe_flow --device-loop (cuChanMgrDev module: device ports, one kernel behind the scan, fixes from the pinned ring,
    the flow thread 3 windows ahead) writes the X-file rows of the host-driven flow (cuChanMgr + cuEKF + XECEFLogger)."""
    import os
    import subprocess
    W, fs, S, K = 30, 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0)
    dat = str(tmp_path / "synthetic_2500kHz.dat")
    iq.tofile(dat)
    ho_path = str(tmp_path / "handoff.csv")
    with open(dpe.workload.HANDOFF_CSV) as f, open(ho_path, "w") as g:
        for line in f:
            g.write("bytes_read,0\n" if line.startswith("bytes_read") else line)
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    rows = {}
    for name, extra in (("host", []), ("dev", ["--device-loop", "--fix-lag", "3"]), ("dev1", ["--device-loop", "--fix-lag", "1"])):
        out = str(tmp_path / ("X_%s.csv" % name))
        subprocess.check_call([exe, "--samples", dat, "--handoff", ho_path, "--out", out, "--iters", str(W), "--grid-dim", "9",
                               "--spacing", "1.0", "--init-delta", "2", "-1", "1", "3"] + extra)
        rows[name] = np.loadtxt(out, delimiter=",")
        assert rows[name].shape == (W, 8)
    assert np.array_equal(rows["dev"], rows["host"]) and np.array_equal(rows["dev1"], rows["host"])     # "%f" rows


def test_cpp_flow_device_loop_with_the_filter_matches_the_host_driven_filter(tmp_path):
    """EnableEKF = true: host/dpe_flow --ekf runs dsp::cuEKF's StepUpdate / StepPredict (cuekf.cu:626-742) on the host in fp64
    (dpe_ekf_*, pinned by fixture O10); --device-loop --ekf runs the same 8 x 8 steps inside the channel manager's measurement
    kernel, one lane per matrix element, the same operations in the same order (dpe_chm_dev_set_ekf).  The logged state x_k|k and,
    through the predicted state that centres the next window's grids, every later fix: identical X-file rows."""
    import os
    import subprocess
    W, fs, S, K = 30, 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0, velocity=np.array([4.0, -2.0, 1.0]))
    dat = str(tmp_path / "synthetic_2500kHz.dat")
    iq.tofile(dat)
    ho_path = str(tmp_path / "handoff.csv")
    with open(dpe.workload.HANDOFF_CSV) as f, open(ho_path, "w") as g:
        for line in f:
            g.write("bytes_read,0\n" if line.startswith("bytes_read") else line)
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    rows = {}
    for name, extra in (("host", ["--ekf"]), ("dev", ["--device-loop", "--ekf", "--fix-lag", "3"]), ("pass", ["--device-loop"])):
        out = str(tmp_path / ("X_%s.csv" % name))
        subprocess.check_call([exe, "--samples", dat, "--handoff", ho_path, "--out", out, "--iters", str(W), "--grid-dim", "9",
                               "--spacing", "1.0", "--init-delta", "2", "-1", "1", "3"] + extra)
        rows[name] = np.loadtxt(out, delimiter=",")
        assert rows[name].shape == (W, 8)
    assert np.array_equal(rows["dev"], rows["host"])               # "%f" rows
    assert not np.array_equal(rows["dev"], rows["pass"])           # ... and the filter really ran (the pass-through rows differ)


@pytest.mark.parametrize("order", ["handles_first", "manager_first", "mixed"])
def test_manager_and_attached_handles_may_be_destroyed_in_any_order(order):
    """The device-resident channel manager parks a time update in the attached BatchCorrScores handle and reads the attached
    BatchCorrManifold's keys: whichever of the three is destroyed first tells the others (dpe_*_hook_set_owner), also with a time
    update still parked -- dsp::Flow::Stop() stops modules in Add order, i.e. the handles BEFORE the manager (flow.cu:168-170).
    After the handles are gone the manager still answers reads, and refuses a step with a message."""
    import torch
    fs, S, K = 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(2, fs, S, K, seed=3, amp=200.0)
    ho = dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV)
    g = dpe.synth.uniform_grid(9, 1.0)
    tg = np.unique(g[:, 3])
    L, B = dpe.pipeline.bank_half_widths(g, g, fs, dpe.engine.carr_fft_len(S))
    x = np.array(ho["X_ECEF"], dtype=np.float64)
    iq_d = torch.from_numpy(iq).to("cuda:0")
    bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, g, g, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcm.Start()
    cm = dpe.engine.ChanMgrDev.from_handoff(ho, S / fs, K, tg)
    cm.attach(bcs, bcm, 4)
    cm.Start(x)
    for w in range(2):
        bcs.UpdatePrepared(iq_d[w], K)
        bcm.UpdatePrepared(bcs.CodeScores, bcs.CarrScores, K)
        cm.step()                      # leaves window w's time update parked in `bcs` for the next stage-1 launch
    first = cm.fix(0)
    assert np.isfinite(first["zVal"]).all()
    if order == "handles_first":
        bcs.Stop(); bcm.Stop()
        second = cm.fix(1, timeout_us=2000000)     # the parked update ran when `bcs` went: it sent window 1's fix
        assert second is not None and np.isclose(second["rxTime"] - first["rxTime"], S / fs, rtol=0, atol=1e-9)
        s, e, win = cm.outputs()
        assert np.isfinite(win["rxTime"][0]) and cm.status == 0
        with pytest.raises(dpe.DpeError, match="attached"):
            cm.step()
        cm.Stop()
    elif order == "manager_first":
        cm.Stop()
        # the handles are ordinary handles again: a host-parameter Update works and publishes its results
        _, cs, ce, bw = dpe.workload.build_windows(1, fs, S, K, seed=3, amp=200.0)
        bcs.Update(iq_d[0], cs[0])
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[:1], ce[0])
        r = bcm.results()[0]
        assert r["posIndex"] >= 0 and np.isfinite(r["zVal"]).all()
        bcm.Stop(); bcs.Stop()
    else:
        bcm.Stop(); cm.Stop(); bcs.Stop()


def test_reference_pair_mode_in_the_device_resident_loop(oracle):
    """referencePair (batchcorrmanifold.cu:1798-1812: floor(idx) / floor(idx + 1) two apart when an fp64 index sits one rounding step
    below 2^m) through dpe_bcm_update_prepared: the prepared blocks hold expansion coefficients only, so the fp64 re-evaluation of the
    affected grid points reads the port arrays of the attached channel manager (handed over at dpe_chm_dev_attach) -- candidates,
    patch and re-derived arg-max on the device, nothing read back.  S / 2 = 4096; uniform grids on the handoff geometry, where the
    branch occurs.  Which points take the branch is a one-ulp property of the inputs, so the device loop is not compared with the
    host-driven loop (their channel managers agree to 1e-12, not to the last bit) but with the FAITHFUL oracle evaluated on the device
    manager's own outputs for each window: ordinary points within the reference's own index noise, the branch's points -- which
    the oracle reports -- within 5e-6, the arg-max the first maximum of the patched scores; and the mode changed something."""
    o = oracle
    fs, S, K, W = 2.5e6, 8192, 6, 4
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=1, amp=200.0)
    ho = dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV)
    g = dpe.synth.uniform_grid(5, 1.0)
    tg = np.zeros(1)
    L, B = dpe.pipeline.bank_half_widths(g, g, fs, dpe.engine.carr_fft_len(S))
    fd, rd, status = dpe.pipeline.run_device_loop(iq, ho, fs, g, g, time_grid=tg, K=K, ring_depth=8, reference_pair=True, keep_scores=True)
    fp, rp, _ = dpe.pipeline.run_device_loop(iq, ho, fs, g, g, time_grid=tg, K=K, ring_depth=8, reference_pair=False, keep_scores=True)
    assert status == 0
    from tests import helpers
    branch = changed = 0
    for w in range(W):
        _, e, win = rd[w]["inputs"]
        sp, _ = o.bcm_pos(e["satState"], rd[w]["codeBank"], S // 2 - L, win["xCurrkk1"][0], g, win["enu2ecef"][0], e["codeFrequency"], e["cpRefTOW"],
                          e["cpElapsedEnd"], e["cpRef"], e["codePhaseEnd"], float(win["rxTime"][0]), fs, S, 1)
        q = o.bcm_pos_quirks()
        got = rd[w]["posScores"]
        assert np.abs(got - sp).max() < helpers.POS_REF_NOISE * sp.max(), w
        if len(q):
            assert np.abs(got[q] - sp[q]).max() < 5e-6 * sp.max(), w
            branch += len(q)
        assert rd[w]["posIndex"] == int(np.argmax(got)) and rd[w]["posScore"] == got.max(), w
        changed += int(np.sum(np.abs(got - rp[w]["posScores"]) > 1e-3 * sp.max()))
    assert branch > 0, "the case must exercise the branch"
    assert changed > 0       # ... and without the mode those points carry the continuous interpolation's value


@pytest.mark.parametrize("couple", [True, False])
def test_device_filter_equals_the_host_filter_with_and_without_velocity_coupling(couple):
    """EnableEKF = true with F = I + T on [i][i + 4] (EKF_MakeDPERandomWalkFMatrix, cuekf.cu:111-143) and with F = I (ekf.py:47).  The
    filter inside the measurement kernel (one lane per matrix element, the structure of F and H exploited, LU on registers with a scalar
    pivot search) is the host filter (dpe_ekf_*, pinned by fixture O10) operation for operation: fed with the measurements the DEVICE loop
    formed (its zVal port, read back window by window), the host filter reproduces the device's x_k|k BIT FOR BIT over the whole run.
    (The two closed loops as wholes agree to ~1e-15 only: the ENU matrix in the measurement comes from either manager's own sin / cos.)"""
    import torch
    W, fs, S, K = 60, 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0, velocity=np.array([4.0, -2.0, 1.0]))
    ho = dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV)
    pos = dpe.synth.uniform_grid(9, 1.0)
    vel = dpe.synth.rand_grid(4, 6561, half=(6.0, 6.0, 6.0, 3.0))
    nfft = dpe.engine.carr_fft_len(S)
    L, B = dpe.pipeline.bank_half_widths(pos, vel, fs, nfft)
    bcs = dpe.engine.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcs.Start()
    bcm = dpe.engine.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, LPower=1, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcm.Start()
    cm = dpe.engine.ChanMgrDev.from_handoff(ho, S / fs, K, (0.0,))
    cm.attach(bcs, bcm, 64)
    x = np.array(ho["X_ECEF"], dtype=np.float64).copy()
    cm.set_ekf(S / fs, x, couple_velocity=couple)
    z_dev = cm.ports()[5]
    iq_d = torch.from_numpy(np.ascontiguousarray(iq)).to("cuda:0")
    cm.Start(x, None)
    host = dpe.engine.cuEKF(x, SampleLength=S / fs, EnableEKF=True, couple_velocity=couple)
    worst = 0.0
    for w in range(W):
        bcs.UpdatePrepared(iq_d[w], K, None)
        bcm.UpdatePrepared(bcs.CodeScores, bcs.CarrScores, K, None)
        cm.step(None)
        r = cm.fix(w)
        assert r["status"] == 0
        z = dpe.engine.d2h(z_dev, 64, np.float64)          # the measurement this window's filter step consumed
        host.Update(z, np.eye(8))
        assert np.array_equal(host.xCurrk1k1, r["zVal"]), (w, np.abs(host.xCurrk1k1 - r["zVal"]).max())
        worst = max(worst, np.abs(r["zVal"][:3] - x[:3]).max())
    assert worst < 50.0
    host.Stop(); cm.Stop(); bcm.Stop(); bcs.Stop()
