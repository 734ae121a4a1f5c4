"""GPU parity tests (run with -m gpu on the MI355X box): HIP path through the C-ABI vs the fp64
oracle on the same seeded inputs, and vs the golden fixtures from the reference's Python twin.

Tolerances (north_star: "within stated fp32 tolerance"), all relative to the peak magnitude of a score bank / the maximum
manifold score of the window:
  * code and Doppler banks, velocity-manifold scores: 2e-6 (TOL) against the fp64 oracle and the PyGNSS fixtures;
  * position-manifold scores: 2e-6 against the oracle evaluated with the EXTENDED-precision (long double) centre index, and
    1e-4 (helpers.POS_REF_NOISE; 3e-4 in the randomised sweep) against the FAITHFUL fp64 oracle and the PyGNSS fixture O7 --
    the reference's own rxTime - pr / C (batchcorrmanifold.cu:1784, rxTime ~ 4e5 s) rounds at 5.8e-11 s = 1.4e-4 samples per
    (point, SV), so the faithful evaluation differs from its own long-double twin by 2.5e-5; 1e-4 is that noise, not fp32's;
  * identical arg-max index (an fp32 tie is the only accepted difference) and identical fix; DC mean and nav-bit bookkeeping
    bit-exact;
  * with banks deliberately narrower than the grid reaches, helpers.assert_parity sets aside at most 16 + pairs / 2000 position
    points and two velocity points per window (pairs within the reference's index noise of the bank edge) -- never when
    posOutOfWindow == 0."""
import ctypes as C

import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe
from tests import helpers

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _bcs_on_fixture(g, L, B):
    import torch
    K = len(g["prn"])
    S = int(g["S"])
    bcs = dpe.BatchCorrScores(float(g["fs"]), samples_per_window=S, lag_half_width=L, bin_half_width=B,
                              max_windows=1, max_channels=K)
    bcs.Start()
    cs = dpe.engine.chan_start_array(g["prn"], g["rc"], g["ri"], g["fc"], g["fi"], g["cp"], g["cp_ref"])
    iq_d = torch.from_numpy(g["iq"]).to("cuda:0")
    bcs.Update(iq_d, cs)
    code, carr = bcs.read_banks()
    info = bcs.read_info()
    nfft = bcs.NumFFTPoints
    _bcs_on_fixture.kernel = bcs.stage1_kernel
    bcs.Stop()
    return code[0], carr[0], info, nfft


@pytest.mark.parametrize("name,L,B", [("o3_handoff_20ms", 8, 48), ("o3_handoff_20ms", 32, 140), ("o3_short_5ms", 4, 30),
                                      ("o12_highrate_4ms", 31, 16), ("o12_highrate_4ms", 64, 16), ("o12_highrate_4ms", 64, 40)])
def test_bcs_vs_reference_fixture(golden, name, L, B):
    """BatchCorrScores banks == pygnss vector_correlate_unfolded windows (fixtures O3; O12 = the same reference function at
    25 Msps: the +-31-lag case runs in bcs_bank_chip2_kernel, +-64 lags with its side chunks from the first chip form, and
    with +-40 bins -- beyond the chip kernels' second-order bound -- in chunks of the boundary-difference kernel)."""
    g = golden(name)
    code, carr, info, nfft = _bcs_on_fixture(g, L, B)
    assert nfft == int(g["C"])
    if name == "o12_highrate_4ms":
        assert _bcs_on_fixture.kernel == ("bcs_bank_chip2_kernel" if B == 16 else "bcs_bank_wide_kernel")
    for k in range(len(g["prn"])):
        rc = g["code"][k][64 - L:64 + L + 1]
        rf = g["carr"][k][256 - B:256 + B + 1]
        assert np.abs(code[k] - rc).max() < TOL * np.abs(g["code"][k]).max()
        assert np.abs(carr[k] - rf).max() < TOL * np.abs(g["carr"][k]).max()


def test_dp_iteration_vs_reference_fixture(golden, oracle):
    """Whole path on the handoff state == one Receiver.dp_track iteration (fixture O7):
    scores, arg-max grid point and the resulting fix."""
    import torch
    g = golden("o7_dp_iteration")
    ho = dpe.handoff.read_handoff(helpers.HANDOFF)
    K, fs, S, C = 8, float(g["fs"]), int(g["S"]), int(g["C"])
    cm = oracle.ChanMgr(ho["prn_list"], ho["rc"], ho["ri"], ho["fc"], ho["fi"], ho["cp"], ho["cp_timestamp"],
                        ho["TOW"], ho["eph"], ho["rxTime"], 0.02)
    pos, vel = dpe.synth.spread_grid()
    tg = np.unique(pos[:, 3])
    X = ho["X_ECEF"]
    batch, R = cm.start(X, X, tg)
    L, B = 8, 48
    bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=K)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcm.Start()
    cs = dpe.engine.chan_start_array(ho["prn_list"], cm.rcStart, cm.riStart, cm.fc, cm.fi, cm.cpElaStart, cm.cpRef)
    ce = dpe.engine.chan_end_array(batch[:, tg.size // 2], cm.rcEnd, cm.fc, cm.fi, cm.cpRefTOW, cm.cpElaEnd, cm.cpRef)
    bw = dpe.engine.bcm_window_array(X[None, :], R[None, :], [cm.rxTime])
    bcs.Update(torch.from_numpy(g["iq"]).to("cuda:0"), cs)
    bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
    r = bcm.results()[0]
    ps, vs = bcm.read_scores()
    assert r["posOutOfWindow"] == 0 and r["velOutOfWindow"] == 0
    # position manifold: the reference's own fp64 index noise (helpers.POS_REF_NOISE) bounds this one
    assert np.abs(ps[0][::97] - g["pos_every97"]).max() < helpers.POS_REF_NOISE * g["pos_every97"].max()
    assert np.abs(vs[0][::97] - g["vel_every97"]).max() < TOL * g["vel_every97"].max()
    assert np.abs(ps[0][g["top_pos_idx"]] - g["top_pos"]).max() < helpers.POS_REF_NOISE * g["top_pos"].max()
    assert np.abs(vs[0][g["top_vel_idx"]] - g["top_vel"]).max() < TOL * g["top_vel"].max()
    assert r["posIndex"] == int(g["argmax_pos"]) and r["velIndex"] == int(g["argmax_vel"])
    assert np.abs((r["zVal"] - X) - g["e"]).max() < 1e-6          # fix within 1 um of the reference's
    bcm.Stop()
    bcs.Stop()


def test_device_resident_ports_match_the_host_form(golden, oracle):
    """dpe_bcs_update_dev / dpe_bcm_update_dev: the channel parameters where the reference keeps them -- device arrays in
    the layout of cuChanMgr's ports (dpeflow.cpp:169-191), the full SatStates batch included -- on the O7 state.  The
    one-block prep kernels derive in fp64 on the device what the host form derives on the host: banks within fp32
    rounding of the host form (libm vs device cos/sin in one rotation constant), scores, arg-max and fix the same, and
    the reference fixture's assertions hold for this form too."""
    import torch
    g = golden("o7_dp_iteration")
    ho = dpe.handoff.read_handoff(helpers.HANDOFF)
    K, fs, S = 8, float(g["fs"]), int(g["S"])
    cm = oracle.ChanMgr(ho["prn_list"], ho["rc"], ho["ri"], ho["fc"], ho["fi"], ho["cp"], ho["cp_timestamp"],
                        ho["TOW"], ho["eph"], ho["rxTime"], 0.02)
    pos, vel = dpe.synth.spread_grid()
    tg = np.unique(pos[:, 3])
    X = ho["X_ECEF"]
    batch, R = cm.start(X, X, tg)
    L, B = 8, 48
    dev = torch.device("cuda:0")
    iq_d = torch.from_numpy(g["iq"]).to(dev)

    def make():
        bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=K)
        bcs.Start()
        bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_channels=K)
        bcm.Start()
        return bcs, bcm

    # host form
    bcs, bcm = make()
    cs = dpe.engine.chan_start_array(ho["prn_list"], cm.rcStart, cm.riStart, cm.fc, cm.fi, cm.cpElaStart, cm.cpRef)
    ce = dpe.engine.chan_end_array(batch[:, tg.size // 2], cm.rcEnd, cm.fc, cm.fi, cm.cpRefTOW, cm.cpElaEnd, cm.cpRef)
    bw = dpe.engine.bcm_window_array(X[None, :], R[None, :], [cm.rxTime])
    bcs.Update(iq_d, cs)
    bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
    r0 = bcm.results()[0]
    code0, carr0 = bcs.read_banks()
    info0 = bcs.read_info()
    ps0, vs0 = bcm.read_scores()
    bcm.Stop(); bcs.Stop()

    # device form: every port array on the device, in the reference's types
    def d(a, dt):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dt)).to(dev)
    keep = dict(rc=d(cm.rcStart, np.float64), ri=d(cm.riStart, np.float64), fc=d(cm.fc, np.float64), fi=d(cm.fi, np.float64),
                ela=d(cm.cpElaStart, np.int32), ref=d(cm.cpRef, np.int32), prn=d(ho["prn_list"], np.uint8),
                x=d(X, np.float64), R=d(np.asarray(R).ravel(), np.float64), sat=d(batch, np.float64), rcE=d(cm.rcEnd, np.float64),
                tow=d(cm.cpRefTOW, np.int32), elaE=d(cm.cpElaEnd, np.int32), ds=d([1], np.int32))
    assert batch.shape == (K, tg.size, 8)
    bcs, bcm = make()
    bcs.UpdateDev(iq_d, K, dict(codePhaseStart=keep["rc"], carrierPhaseStart=keep["ri"], codeFrequency=keep["fc"],
                                carrierFrequency=keep["fi"], cpElapsedStart=keep["ela"], cpReference=keep["ref"], validPRNs=keep["prn"]))
    bcm.UpdateDev(bcs.CodeScores, bcs.CarrScores, K,
                  dict(xCurrkk1=keep["x"], enu2ecef=keep["R"], satStates=keep["sat"], codePhaseEnd=keep["rcE"], codeFrequency=keep["fc"],
                       carrierFrequency=keep["fi"], cpRefTOW=keep["tow"], cpElapsedEnd=keep["elaE"], cpRef=keep["ref"], dopplerSign=keep["ds"]),
                  tg.size, cm.rxTime)
    r1 = bcm.results()[0]
    assert bcs.dev_status() == 0
    code1, carr1 = bcs.read_banks()
    info1 = bcs.read_info()
    ps1, vs1 = bcm.read_scores()
    for k in range(K):
        assert np.abs(code1[0][k] - code0[0][k]).max() < 2e-7 * np.abs(code0[0][k]).max()
        assert np.abs(carr1[0][k] - carr0[0][k]).max() < 2e-7 * np.abs(carr0[0][k]).max()
    assert np.array_equal(info1[0], info0[0]) and np.array_equal(info1[1], info0[1]) and np.array_equal(info1[2], info0[2])
    assert np.abs(ps1[0] - ps0[0]).max() < 1e-6 * ps0[0].max() and np.abs(vs1[0] - vs0[0]).max() < 1e-6 * vs0[0].max()
    assert r1["posIndex"] == r0["posIndex"] == int(g["argmax_pos"]) and r1["velIndex"] == r0["velIndex"] == int(g["argmax_vel"])
    assert np.array_equal(r1["zVal"], r0["zVal"]) and r1["posOutOfWindow"] == 0 and r1["velOutOfWindow"] == 0
    assert np.abs(vs1[0][::97] - g["vel_every97"]).max() < TOL * g["vel_every97"].max()
    assert np.abs(ps1[0][::97] - g["pos_every97"]).max() < helpers.POS_REF_NOISE * g["pos_every97"].max()
    # a PRN outside 1..37 on the device is clamped and reported, never an out-of-table read
    keep["prn"][0] = 99
    bcs.UpdateDev(iq_d, K, dict(codePhaseStart=keep["rc"], carrierPhaseStart=keep["ri"], codeFrequency=keep["fc"],
                                carrierFrequency=keep["fi"], cpElapsedStart=keep["ela"], cpReference=keep["ref"], validPRNs=keep["prn"]))
    assert bcs.dev_status() & 1
    # ... and a negative code phase / non-positive code frequency is flagged, with nominal values in the kernels' place (the
    # chip-table index of a negative phase would fall in front of the table) -- ADVICE r3
    keep["prn"][0] = int(ho["prn_list"][0])
    keep["rc"][1] = -3.25
    keep["fc"][2] = 0.0
    bcs.UpdateDev(iq_d, K, dict(codePhaseStart=keep["rc"], carrierPhaseStart=keep["ri"], codeFrequency=keep["fc"],
                                carrierFrequency=keep["fi"], cpElapsedStart=keep["ela"], cpReference=keep["ref"], validPRNs=keep["prn"]))
    assert bcs.dev_status() == 2
    code2, _ = bcs.read_banks()
    assert np.isfinite(code2).all()
    for k in (0, 3, 4, 5, 6, 7):     # the channels with sane inputs are what they were
        assert np.abs(code2[0][k] - code0[0][k]).max() < 2e-7 * np.abs(code0[0][k]).max()
    bcm.Stop(); bcs.Stop()


@pytest.mark.parametrize("kw,L,B", [
    (dict(seed=1, S=12500, K=4, G=5000, amp=200.0), 8, 32),                       # ragged grid (not /1024)
    (dict(seed=2, S=50000, K=8, G=20000, amp=48.0), 8, 48),                       # 45 dB-Hz
    (dict(seed=3, S=50000, K=8, G=4096, amp=200.0, W=3), 4, 40),                  # batch of 3 windows
    (dict(seed=4, S=50000, K=8, G=10000, amp=200.0, center_offset=(3.0, -2.0, 4.0, 5.0)), 8, 48),
    (dict(seed=5, S=50000, K=8, G=6561, amp=200.0, grid="uniform"), 4, 60),       # reference default spacing
    (dict(seed=6, S=50000, K=1, G=1024, amp=200.0), 8, 48),                       # single SV
])
def test_path_vs_oracle(kw, L, B):
    case = helpers.make_case(**kw)
    out = helpers.run_gpu(case, L, B)
    ref = helpers.run_oracle(case, L, B)
    worst = helpers.assert_parity(out, ref, tol=TOL)
    print("worst rel err", worst)


def test_config_h_25msps_12sv():
    """BASELINE.json configs[2]: 25 Msps x 20 ms (S = 500000, beyond the reference's unsigned-short
    lengths), 12 SVs, 1e5-point random ENU-dt grid -> +-31 lags, +-16 bins of the 4194304-point FFT."""
    cfg = dpe.workload.CONFIG_H
    case = helpers.make_case(seed=21, fs=cfg["fs"], S=cfg["S"], K=cfg["K"], G=cfg["G"], amp=60.0)
    assert case["C"] == 4194304
    out = helpers.run_gpu(case, cfg["L"], cfg["B"])
    ref = helpers.run_oracle(case, cfg["L"], cfg["B"])
    worst = helpers.assert_parity(out, ref, tol=TOL)
    assert ref["res"][0]["posOutOfWindow"] == 0 and ref["res"][0]["velOutOfWindow"] == 0
    print("config H worst rel err", worst)


@pytest.mark.parametrize("lpower", [2, 3])
def test_lpower(lpower):
    """LPower param (batchcorrmanifold.cu:2290): |.|^2 has its own variant (no square root); any other power goes
    through the generic powf variant (fp32 powf: tolerance 1e-5 there)."""
    case = helpers.make_case(seed=7, S=12500, K=4, G=3000, amp=200.0)
    out = helpers.run_gpu(case, 8, 32, lpower=lpower)
    ref = helpers.run_oracle(case, 8, 32, lpower=lpower)
    helpers.assert_parity(out, ref, tol=TOL if lpower == 2 else 1e-5)


def test_out_of_window_points_are_counted():
    """Grid reaching beyond the lag bank: pairs are dropped (score 0) and counted, identically in
    the oracle (the reference leaves this undefined, batchcorrmanifold.cu:1795-1804)."""
    case = helpers.make_case(seed=8, S=12500, K=4, G=2000, amp=200.0)
    case["pos"][:, 3] *= 20.0   # +-2.6 km clock offsets -> ~+-22 samples
    out = helpers.run_gpu(case, 4, 32)
    ref = helpers.run_oracle(case, 4, 32)
    assert ref["res"][0]["posOutOfWindow"] > 0
    helpers.assert_parity(out, ref, tol=TOL)


def test_dense_export_matches_reference_layout(golden):
    import torch
    g = golden("o3_short_5ms")
    K, S, C = len(g["prn"]), int(g["S"]), int(g["C"])
    L, B = 4, 30
    bcs = dpe.BatchCorrScores(float(g["fs"]), samples_per_window=S, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcs.Start()
    cs = dpe.engine.chan_start_array(g["prn"], g["rc"], g["ri"], g["fc"], g["fi"], g["cp"], g["cp_ref"])
    bcs.Update(torch.from_numpy(g["iq"]).to("cuda:0"), cs)
    code_d = torch.empty(K * S, dtype=torch.complex128, device="cuda:0")
    carr_d = torch.empty(K * C, dtype=torch.complex128, device="cuda:0")
    bcs.export_dense(0, code_d, carr_d)
    torch.cuda.synchronize()
    code, carr = code_d.cpu().numpy().reshape(K, S), carr_d.cpu().numpy().reshape(K, C)
    for k in range(K):
        assert np.abs(code[k, S // 2 - L:S // 2 + L + 1] - g["code"][k][64 - L:64 + L + 1]).max() < TOL * np.abs(g["code"][k]).max()
        assert np.abs(carr[k, C // 2 - B:C // 2 + B + 1] - g["carr"][k][256 - B:256 + B + 1]).max() < TOL * np.abs(g["carr"][k]).max()
        assert np.count_nonzero(code[k]) <= 2 * L + 1 and np.count_nonzero(carr[k]) <= 2 * B + 1
    bcs.Stop()


def test_pos_scores_port_in_the_reference_type():
    """dpe_bcm_export_scores_f64: the PosScores port as the reference declares it (DOUBLE_t, GRID: one dense fp64 row,
    batchcorrmanifold.cu:2300) for a chosen window of a batch whose fp32 rows are pitched."""
    import torch
    case = helpers.make_case(seed=31, S=12500, K=4, G=3001, amp=200.0, W=3)
    iq, cs, ce, bw = helpers.pack_gpu_inputs(case)
    L, B = 8, 32
    bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=case["S"], lag_half_width=L, bin_half_width=B, max_windows=3, max_channels=4)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(case["fs"], case["S"], bcs.NumFFTPoints, case["pos"], case["vel"], lag_half_width=L, bin_half_width=B,
                                max_windows=3, max_channels=4)
    bcm.Start()
    bcs.Update(torch.from_numpy(iq).to("cuda:0"), cs)
    bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
    ps, vs = bcm.read_scores()
    assert bcm.PosScoresPitch > 3001
    p64 = torch.full((3001,), -1.0, dtype=torch.float64, device="cuda:0")
    v64 = torch.full((3001,), -1.0, dtype=torch.float64, device="cuda:0")
    bcm.export_scores_f64(2, p64, v64)
    torch.cuda.synchronize()
    assert np.array_equal(p64.cpu().numpy(), ps[2].astype(np.float64)) and np.array_equal(v64.cpu().numpy(), vs[2].astype(np.float64))
    with pytest.raises(dpe.DpeError):
        bcm.export_scores_f64(3, p64, None)
    bcm.Stop(); bcs.Stop()


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_sharded_grid_equals_full_grid():
    """Two BatchCorrManifold instances on the two halves of a grid (index offsets as in the multi-GPU
    path) + integer max of their packed keys == one instance on the whole grid."""
    import torch
    case = helpers.make_case(seed=9, S=12500, K=4, G=5001, amp=200.0, W=2)
    iq, cs, ce, bw = helpers.pack_gpu_inputs(case)
    L, B = 8, 32
    bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=case["S"], lag_half_width=L, bin_half_width=B, max_windows=2, max_channels=4)
    bcs.Start()
    bcs.Update(torch.from_numpy(iq).to("cuda:0"), cs)

    def run(pos, vel, po, vo):
        m = dpe.BatchCorrManifold(case["fs"], case["S"], bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B,
                                  max_windows=2, max_channels=4, pos_index_offset=po, vel_index_offset=vo)
        m.Start()
        m.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
        torch.cuda.synchronize()
        keys = dpe.engine.d2h(m.Keys, 2 * 2 * 8, np.uint64).reshape(2, 2)
        return m, keys

    full, kf = run(case["pos"], case["vel"], 0, 0)
    rf = full.results()
    b0, e0 = dpe.sharding.shard_range(5001, 0, 2)
    b1, e1 = dpe.sharding.shard_range(5001, 1, 2)
    m0, k0 = run(case["pos"][b0:e0], case["vel"][b0:e0], b0, b0)
    m1, k1 = run(case["pos"][b1:e1], case["vel"][b1:e1], b1, b1)
    merged = np.maximum(k0, k1)
    assert np.array_equal(merged, kf)
    rm = m0.results_from_keys(merged, case["pos"], case["vel"])
    for w in range(2):
        assert rm[w]["posIndex"] == rf[w]["posIndex"] and rm[w]["velIndex"] == rf[w]["velIndex"]
        assert np.array_equal(rm[w]["zVal"], rf[w]["zVal"])
    # a key of 0 (no valid score / a key set that was never reduced) or an index beyond the global grids is an error
    bad = merged.copy()
    bad[1, 0] = 0
    with pytest.raises(dpe.DpeError, match="no valid score"):
        m0.results_from_keys(bad, case["pos"], case["vel"])
    with pytest.raises(dpe.DpeError, match="outside the global grids"):
        m0.results_from_keys(merged, case["pos"][:10], case["vel"])
    for m in (full, m0, m1):
        m.Stop()
    bcs.Stop()


def test_closed_loop_poll_and_stream_wait_modes_agree(monkeypatch):
    """dpe_bcm_results either polls the sequence word the scan's last block writes behind the results (default for single
    windows) or waits for the stream (DPE_BCM_NO_POLL=1, read at create): same fixes, window by window."""
    W, fs, S, K = 12, 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=8, amp=200.0)
    ho = dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV)
    g9 = dpe.synth.uniform_grid(9, 1.0)
    polled, _ = dpe.pipeline.run_closed_loop(iq, ho, fs, g9, g9, time_grid=np.unique(g9[:, 3]))
    monkeypatch.setenv("DPE_BCM_NO_POLL", "1")
    waited, _ = dpe.pipeline.run_closed_loop(iq, ho, fs, g9, g9, time_grid=np.unique(g9[:, 3]))
    assert np.array_equal(polled, waited)
    assert np.abs(polled[:, :3] - ho["X_ECEF"][:3]).max() < 1.0


def test_cpp_flow_matches_python_closed_loop(tmp_path):
    """host/dpe_flow (C++ modules wired as DPEFlow::LoadFlow) on a synthetic sample file == the Python
    closed loop on the same library; and the fix stays on the true position (static receiver)."""
    import os
    import subprocess
    W, fs, S, K = 6, 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0)
    dat = str(tmp_path / "synthetic_2500kHz.dat")
    iq.tofile(dat)
    ho_path = str(tmp_path / "handoff.csv")
    with open(dpe.workload.HANDOFF_CSV) as f, open(ho_path, "w") as g:
        for line in f:
            g.write("bytes_read,0\n" if line.startswith("bytes_read") else line)
    out = str(tmp_path / "X.csv")
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    subprocess.check_call([exe, "--samples", dat, "--handoff", ho_path, "--out", out, "--iters", str(W), "--grid-dim", "9",
                           "--spacing", "1.0"])
    rows = np.loadtxt(out, delimiter=",")
    assert rows.shape == (W, 8)
    ho = dpe.handoff.read_handoff(ho_path)
    g9 = dpe.synth.uniform_grid(9, 1.0)
    fixes, res = dpe.pipeline.run_closed_loop(iq, ho, fs, g9, g9, time_grid=np.unique(g9[:, 3]))
    assert np.abs(rows - fixes).max() < 1e-6                       # %f rows vs doubles
    assert np.abs(fixes[:, :3] - ho["X_ECEF"][:3]).max() < 1.0     # position fix within 1 m of truth
    assert all(r["posOutOfWindow"] == 0 and r["velOutOfWindow"] == 0 for r in res)
    # RINEXFilename set (what the reference's DPInit requires, dpinit.cpp:130-144): ephemerides from the nav file,
    # closest toe; the same broadcast values as the handoff rows to 12 digits -> the same fixes
    rnx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "o11_nist1860_excerpt.18n")
    out3 = str(tmp_path / "X_rinex.csv")
    subprocess.check_call([exe, "--samples", dat, "--handoff", ho_path, "--out", out3, "--iters", str(W), "--grid-dim", "9",
                           "--spacing", "1.0", "--rinex", rnx])
    assert np.abs(np.loadtxt(out3, delimiter=",") - rows).max() < 1e-5
    # EnableEKF=true (the reference ships it off, dpeflow.cpp:90): the fix goes through the 8-state filter
    # (dpe_ekf_*, StepUpdate + StepPredict per window) before it is fed back; a static receiver stays on the truth
    out2 = str(tmp_path / "X_ekf.csv")
    subprocess.check_call([exe, "--samples", dat, "--handoff", ho_path, "--out", out2, "--iters", str(W), "--grid-dim", "9",
                           "--spacing", "1.0", "--ekf"])
    rows2 = np.loadtxt(out2, delimiter=",")
    assert rows2.shape == (W, 8)
    assert np.abs(rows2[:, :3] - ho["X_ECEF"][:3]).max() < 1.5
    assert np.abs(rows2[:, 4:7]).max() < 1.5                        # velocity estimate of a static receiver


@pytest.mark.parametrize("half,expect_wide", [((4.0, 4.0, 4.0, 4.0), False), ((1500.0, 1500.0, 1500.0, 1500.0), True)])
def test_cpp_flow_with_a_loaded_rngrid_csv(tmp_path, half, expect_wide):
    """LoadPosGrid / LoadPosGridFilename (batchcorrmanifold.cu:2422-2448; dpeflow.cpp:133): a 9^4-row rngrid-format CSV
    replaces the built position grid.  dpe_flow --load-grid == the Python closed loop on the same grid, and a file with
    the wrong row count is refused (the reference would overrun its buffer).  The second case spreads the grid over
    +-1500 m: +-37 code lags at 2.5 Msps, beyond the single 65-lag chunk."""
    import os
    import subprocess
    W, fs, S, K = 4, 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=6, amp=200.0)
    dat = str(tmp_path / "synthetic_2500kHz.dat")
    iq.tofile(dat)
    ho_path = str(tmp_path / "handoff.csv")
    with open(dpe.workload.HANDOFF_CSV) as f, open(ho_path, "w") as g:
        for line in f:
            g.write("bytes_read,0\n" if line.startswith("bytes_read") else line)
    rg = dpe.synth.rand_grid(77, 9 ** 4, half=half)
    rg[0] = 0.0                                                    # keep the truth on the grid
    csv = str(tmp_path / "rngrid_test.csv")
    dpe.synth.write_grid_csv(csv, rg)
    out = str(tmp_path / "X.csv")
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    r = subprocess.run([exe, "--samples", dat, "--handoff", ho_path, "--out", out, "--iters", str(W), "--grid-dim", "9",
                        "--spacing", "1.0", "--load-grid", csv], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = np.loadtxt(out, delimiter=",")
    assert rows.shape == (W, 8)
    ho = dpe.handoff.read_handoff(ho_path)
    g9 = dpe.synth.uniform_grid(9, 1.0)                            # the built grid: velocity manifold and TimeGrid
    L, _ = dpe.pipeline.bank_half_widths(rg, g9, fs, dpe.engine.carr_fft_len(S))
    assert (L > 32) == expect_wide and ("L=%d," % L) in r.stderr
    fixes, res = dpe.pipeline.run_closed_loop(iq, ho, fs, rg, g9, time_grid=np.unique(g9[:, 3]))
    assert np.abs(rows - fixes).max() < 1e-6                       # %f rows vs doubles: the same grid points
    assert all(r_["posOutOfWindow"] == 0 for r_ in res)
    # the loaded grid was really used: the fixes are points of the CSV grid about the fed-back centre
    assert np.abs(fixes[:, :3] - ho["X_ECEF"][:3]).max() < (1.0 if not expect_wide else 250.0)
    short = str(tmp_path / "short.csv")
    dpe.synth.write_grid_csv(short, rg[:-1])
    r2 = subprocess.run([exe, "--samples", dat, "--handoff", ho_path, "--out", out, "--iters", "1", "--grid-dim", "9",
                         "--load-grid", short], capture_output=True, text=True)
    assert r2.returncode == 1 and "needs 9^4 rows" in r2.stderr


@pytest.mark.parametrize("kw,L,B", [
    (dict(seed=31, S=12502, K=3, G=1, amp=200.0), 4, 24),        # one grid point; S not a multiple of 4 (scalar loads)
    (dict(seed=32, S=12500, K=2, G=1023, amp=200.0), 4, 24),     # one short of a 1024-point tile
    (dict(seed=33, S=12500, K=2, G=1025, amp=200.0, W=2), 4, 24),  # one over
    (dict(seed=34, S=10240, K=5, G=2048, amp=100.0, W=2), 16, 24),  # S = 40 exact sub-tiles, LH=16 kernel
])
def test_ragged_sizes(kw, L, B):
    case = helpers.make_case(**kw)
    out = helpers.run_gpu(case, L, B)
    ref = helpers.run_oracle(case, L, B)
    helpers.assert_parity(out, ref, tol=TOL)


def test_non_integer_ns_sampling_period():
    """fs = 2.046 MHz (period 488.76 ns): the reference's 1 ns rounding of the sample times
    (BCS_GenTimeIdcs, batchcorrscores.cu:191-193) moves chip boundaries; the kernels then read the
    reference's own time table instead of n/fs."""
    case = helpers.make_case(seed=41, fs=2.046e6, S=40920, K=6, G=3000, amp=200.0)
    out = helpers.run_gpu(case, 8, 40)
    ref = helpers.run_oracle(case, 8, 40)
    helpers.assert_parity(out, ref, tol=TOL)


def test_max_channels_37():
    """CONST_PRN_MAX = 37 tracked SVs (the reference's allocation stride; PRN 37 itself is generated
    here although BCS_GenCACode never writes it, batchcorrscores.cu:127)."""
    case = helpers.make_case(seed=35, S=12500, K=37, G=700, amp=60.0)
    assert sorted(case["prn"])[-1] == 37
    out = helpers.run_gpu(case, 8, 24)
    ref = helpers.run_oracle(case, 8, 24)
    helpers.assert_parity(out, ref, tol=TOL)


def test_37_channels_with_wide_banks_use_the_compact_lds_layout():
    """37 channels x +-140 Doppler bins: 37 x 281 sixteen-byte bank entries (166 KB) do not fit the 160 KB LDS of a CU;
    the scan then keeps 12-byte entries (slower variant, same arithmetic).  S = 50000 so that the 524288-point carrier
    transform takes +-140 bins within the moment expansion (batchcorrmanifold.cu:1861-1963 reads any bin)."""
    case = helpers.make_case(seed=36, S=50000, K=37, G=1500, amp=60.0)
    out = helpers.run_gpu(case, 8, 140)
    ref = helpers.run_oracle(case, 8, 140)
    helpers.assert_parity(out, ref, tol=TOL)
    # beyond the compact layout as well: refused with a message
    with pytest.raises(dpe.DpeError, match="LDS"):
        m = dpe.BatchCorrManifold(case["fs"], case["S"], dpe.engine.carr_fft_len(case["S"]), case["pos"], case["vel"],
                                  lag_half_width=8, bin_half_width=180, max_channels=37)
        m.Start()


def test_window_stride_and_unaligned_base():
    """Windows separated by padding (stride > S) and a sample pointer that is only 4-byte aligned."""
    import torch
    case = helpers.make_case(seed=36, S=12500, K=4, G=600, amp=200.0, W=2)
    iq, cs, ce, bw = helpers.pack_gpu_inputs(case)
    S, stride = case["S"], case["S"] + 37
    buf = np.zeros(1 + 2 * 2 * stride, dtype=np.int16)        # +1 int16 pair offset -> base % 16 == 4
    base = 2
    for w in range(2):
        buf[base + 2 * stride * w: base + 2 * stride * w + 2 * S] = iq[w]
    d = torch.from_numpy(buf).to("cuda:0")
    bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=S, lag_half_width=8, bin_half_width=32, max_windows=2, max_channels=4)
    bcs.Start()
    bcs.Update(d.data_ptr() + 2 * base, cs, window_stride=stride)
    code, carr = bcs.read_banks()
    ref = helpers.run_oracle(case, 8, 32)
    for w in range(2):
        assert np.abs(code[w] - ref["code"][w]).max() < TOL * np.abs(ref["code"][w]).max()
        assert np.abs(carr[w] - ref["carr"][w]).max() < TOL * np.abs(ref["carr"][w]).max()
    bcs.Stop()


def test_argument_errors_are_reported():
    """0 / -1 status with a [Module]-tagged message; nothing is computed on bad input."""
    import torch
    e = dpe.engine
    with pytest.raises(dpe.DpeError, match="lagHalfWidth"):          # (L = 400 is served by the FFT path; half the window is not)
        dpe.BatchCorrScores(2.5e6, samples_per_window=50000, lag_half_width=25000).Start()
    with pytest.raises(dpe.DpeError, match="binHalfWidth"):
        dpe.BatchCorrScores(2.5e6, samples_per_window=50000, bin_half_width=262144).Start()
    bcs = dpe.BatchCorrScores(2.5e6, samples_per_window=50000, max_channels=4)
    bcs.Start()
    iq = torch.zeros(100000, dtype=torch.int16, device="cuda:0")
    good = e.chan_start_array([2], [1.0], [0.0], [1.023e6], [0.0], [0], [0])
    with pytest.raises(dpe.DpeError, match="nChan"):
        bcs.Update(iq, np.repeat(good, 5))
    bad = good.copy(); bad["prn"] = 38
    with pytest.raises(dpe.DpeError, match="PRN"):
        bcs.Update(iq, bad)
    bcs.Update(iq, good)          # all-zero samples: finite, zero banks
    code, carr = bcs.read_banks()
    assert np.all(code == 0) and np.all(carr == 0)
    g = dpe.synth.rand_grid(1, 100)
    with pytest.raises(dpe.DpeError, match="3 km"):
        dpe.BatchCorrManifold(2.5e6, 50000, bcs.NumFFTPoints, g * 100.0, g).Start()
    with pytest.raises(dpe.DpeError, match="even"):
        dpe.BatchCorrManifold(2.5e6, 49999, bcs.NumFFTPoints, g, g).Start()
    bcs.Stop()
    with pytest.raises(dpe.DpeError, match="not initialized"):
        bcs.Update(iq, good)


def test_batch_invariance_and_run_to_run_determinism():
    """Size-independent properties at the full reference shape (S = 50000, 8 SVs, 25^4-point grids):
    a batch of W windows gives bit-identical banks, scores and keys to W single-window calls (for batch
    sizes that share the same tile partition of the bank kernel; larger batches regroup the fp32 partial
    sums), and a
    repeated call reproduces every bit (no float atomics anywhere on the path)."""
    import torch
    cfg = dpe.workload.CONFIG_R
    W = 3
    iq, cs, ce, bw = dpe.workload.build_windows(W, cfg["fs"], cfg["S"], cfg["K"], seed=17, amp=cfg["amp"])
    _, _, pos, vel, _ = dpe.workload.build_grids(cfg["G"])
    L, B = cfg["L"], cfg["B"]
    iq_d = torch.from_numpy(iq).to("cuda:0")

    def run(wsel, maxw):
        bcs = dpe.BatchCorrScores(cfg["fs"], samples_per_window=cfg["S"], lag_half_width=L, bin_half_width=B,
                                  max_windows=maxw, max_channels=cfg["K"])
        bcs.Start()
        bcm = dpe.BatchCorrManifold(cfg["fs"], cfg["S"], bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B,
                                    max_windows=maxw, max_channels=cfg["K"])
        bcm.Start()
        bcs.Update(iq_d[wsel], cs[wsel])
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[wsel], ce[wsel])
        res = bcm.results()
        code, carr = bcs.read_banks()
        ps, vs = bcm.read_scores()
        bcm.Stop(); bcs.Stop()
        return code, carr, ps, vs, res

    a = run(slice(0, W), W)
    b = run(slice(0, W), W)
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)                                   # run-to-run: every bit
    for w in range(W):
        s = run(slice(w, w + 1), 1)
        assert np.array_equal(s[0][0], a[0][w]) and np.array_equal(s[1][0], a[1][w])
        assert np.array_equal(s[2][0], a[2][w]) and np.array_equal(s[3][0], a[3][w])
        assert s[4][0]["posIndex"] == a[4][w]["posIndex"] and s[4][0]["velIndex"] == a[4][w]["velIndex"]
        assert np.array_equal(s[4][0]["zVal"], a[4][w]["zVal"])
        assert a[4][w]["posOutOfWindow"] == 0 and a[4][w]["velOutOfWindow"] == 0
    # arg-max consistency with the materialised scores (first maximum)
    for w in range(W):
        assert a[4][w]["posIndex"] == int(np.argmax(a[2][w])) and a[4][w]["velIndex"] == int(np.argmax(a[3][w]))


def test_graph_replay_equals_eager_launches():
    """dpe_*_set_graph: an Update replayed as one hipGraph (created stream, parameters changing every
    window, two alternating sample buffers as with the SampleBlock ring) reproduces the eager path bit
    for bit; on the null stream the flag silently keeps eager launches."""
    import torch
    cfg = dpe.workload.CONFIG_R
    W = 6
    iq, cs, ce, bw = dpe.workload.build_windows(W, cfg["fs"], cfg["S"], cfg["K"], seed=23, amp=cfg["amp"])
    _, _, pos, vel, _ = dpe.workload.build_grids(6561)
    L, B = cfg["L"], cfg["B"]
    iq_d = torch.from_numpy(iq).to("cuda:0")
    ring = [torch.empty_like(iq_d[0]) for _ in range(2)]

    def run(graph, stream):
        bcs = dpe.BatchCorrScores(cfg["fs"], samples_per_window=cfg["S"], lag_half_width=L, bin_half_width=B,
                                  max_windows=1, max_channels=cfg["K"])
        bcs.Start()
        bcm = dpe.BatchCorrManifold(cfg["fs"], cfg["S"], bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B,
                                    max_windows=1, max_channels=cfg["K"])
        bcm.Start()
        bcs.set_graph(graph); bcm.set_graph(graph)
        out = []
        for w in range(W):
            slot = ring[w % 2]
            slot.copy_(iq_d[w]); torch.cuda.synchronize()
            bcs.Update(slot, cs[w], stream=stream)
            bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[w:w + 1], ce[w], stream=stream)
            r = bcm.results(stream=stream)[0]
            code, carr = bcs.read_banks(stream=stream)
            ps, vs = bcm.read_scores(stream=stream)
            out.append((code.copy(), carr.copy(), ps.copy(), vs.copy(), r))
        bcm.Stop(); bcs.Stop()
        return out

    st = dpe.engine.Stream()
    eager = run(False, st)
    for variant in (run(True, st), run(True, None)):
        for a, b in zip(eager, variant):
            for x, y in zip(a[:4], b[:4]):
                assert np.array_equal(x, y)
            assert a[4]["posIndex"] == b[4]["posIndex"] and a[4]["velIndex"] == b[4]["velIndex"]
            assert np.array_equal(a[4]["zVal"], b[4]["zVal"])
    st.close()
    # the windows differ, so identical results would mean a stale replay
    assert not np.array_equal(eager[0][0], eager[1][0])


def test_key_sets_stay_clean_across_changing_batch_sizes():
    """The arg-max keys / out-of-window counters live in two alternating device sets, each cleared by the
    following Update's position scan.  Repeated Updates with batch sizes 3, 1, 3, 2, 3 on ONE handle must
    give, for every window, exactly what a fresh handle gives (no stale maximum from an earlier call)."""
    import torch
    case = helpers.make_case(seed=31, S=12500, K=5, G=4097, amp=200.0, W=3)
    iq, cs, ce, bw = helpers.pack_gpu_inputs(case)
    L, B = 8, 32
    iq_d = torch.from_numpy(iq).to("cuda:0")

    def handles():
        bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=case["S"], lag_half_width=L, bin_half_width=B, max_windows=3, max_channels=5)
        bcs.Start()
        bcm = dpe.BatchCorrManifold(case["fs"], case["S"], bcs.NumFFTPoints, case["pos"], case["vel"], lag_half_width=L,
                                    bin_half_width=B, max_windows=3, max_channels=5)
        bcm.Start()
        return bcs, bcm

    def update(bcs, bcm, n):
        bcs.Update(iq_d[:n], cs[:n])
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[:n], ce[:n])
        return [(r["posIndex"], r["velIndex"], r["posScore"], r["velScore"], r["posOutOfWindow"], r["velOutOfWindow"])
                for r in bcm.results()]

    bcs, bcm = handles()
    seq = [update(bcs, bcm, n) for n in (3, 1, 3, 2, 3)]
    bcm.Stop(); bcs.Stop()
    for n, got in zip((3, 1, 3, 2, 3), seq):
        b2, m2 = handles()
        assert got == update(b2, m2, n)
        m2.Stop(); b2.Stop()
    assert seq[0] == seq[2] == seq[4] and seq[1] == seq[0][:1] and seq[3] == seq[0][:2]


@pytest.mark.parametrize("L,B,W,fs,S", [(4, 20, 24, 2.5e6, 50000), (8, 40, 24, 2.5e6, 50000), (4, 12, 128, 2.046e6, 8184)])
def test_batch_bank_kernel_matches_single_window_kernel(L, B, W, fs, S):
    """Batches large enough to fill the chip run stage 1 through bcs_bank16_kernel (16 samples per lane);
    single windows use bcs_bank_kernel (4 per lane).  Same arithmetic per sample, different grouping of the
    fp32 sums: banks agree to 1e-6 of the peak (each is within 3e-7 of the fp64 oracle, see the fixture
    tests), DC mean / nav-bit index / replica choice are identical, and so is the fix.  S = 50000 is not a
    multiple of the 1024-sample pass, every window holds a nav-bit edge, B = 40 selects the 6-moment variant, and
    2.046 Msps (sampling period not a whole number of ns) the time-table variant."""
    import torch
    cfg = dpe.workload.CONFIG_R
    cfg = dict(cfg, fs=fs, S=S)
    iq, cs, ce, bw = dpe.workload.build_windows(W, cfg["fs"], cfg["S"], cfg["K"], seed=41, amp=cfg["amp"])
    _, _, pos, vel, _ = dpe.workload.build_grids(6561)
    iq_d = torch.from_numpy(iq).to("cuda:0")

    def run(wsel, maxw):
        bcs = dpe.BatchCorrScores(cfg["fs"], samples_per_window=cfg["S"], lag_half_width=L, bin_half_width=B,
                                  max_windows=maxw, max_channels=cfg["K"])
        bcs.Start()
        bcm = dpe.BatchCorrManifold(cfg["fs"], cfg["S"], bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B,
                                    max_windows=maxw, max_channels=cfg["K"])
        bcm.Start()
        bcs.Update(iq_d[wsel], cs[wsel])
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[wsel], ce[wsel])
        res = bcm.results()
        code, carr = bcs.read_banks()
        info = bcs.read_info()
        bcm.Stop(); bcs.Stop()
        return code, carr, info, res

    code, carr, info, res = run(slice(0, W), W)
    for w in (0, 7, W - 1):
        c1, f1, i1, r1 = run(slice(w, w + 1), 1)
        assert np.abs(code[w] - c1[0]).max() <= 1e-6 * np.abs(c1[0]).max()
        assert np.abs(carr[w] - f1[0]).max() <= 1e-6 * np.abs(f1[0]).max()
        assert np.array_equal(info[0][w], i1[0][0]) and np.array_equal(info[1][w], i1[1][0]) and info[2][w] == i1[2][0]
        assert res[w]["posIndex"] == r1[0]["posIndex"] and res[w]["velIndex"] == r1[0]["velIndex"]


def test_score_rows_start_on_128_byte_lines():
    """dpe_bcm_scores / dpe_bcm_scores_pitch: one row of gridSize floats per window, `pitch` floats apart, pitch = the grid
    size rounded up to a multiple of 32 (aligned rows make a wave's 64 scores two whole lines).  The floats
    between a row's end and the next row are never written, and row w holds window w's scores."""
    import torch
    cfg = dpe.workload.CONFIG_R
    W, L, B, G = 3, 4, 20, 1000
    iq, cs, ce, bw = dpe.workload.build_windows(W, cfg["fs"], cfg["S"], cfg["K"], seed=47, amp=cfg["amp"])
    _, _, pos, vel, _ = dpe.workload.build_grids(G)
    bcs = dpe.BatchCorrScores(cfg["fs"], samples_per_window=cfg["S"], lag_half_width=L, bin_half_width=B, max_windows=W,
                              max_channels=cfg["K"])
    bcs.Start()
    bcm = dpe.BatchCorrManifold(cfg["fs"], cfg["S"], bcs.NumFFTPoints, pos, vel[:777], lag_half_width=L, bin_half_width=B,
                                max_windows=W, max_channels=cfg["K"])
    bcm.Start()
    assert bcm.PosScoresPitch == 1024 and bcm.VelScoresPitch == 800 and bcm.PosScores % 128 == 0 and bcm.VelScores % 128 == 0
    fill = np.float32(-7.0)
    # mark the whole buffers, run, and look at what was written
    for ptr, n in ((bcm.PosScores, W * 1024), (bcm.VelScores, W * 800)):
        src = np.full(n, fill, np.float32)
        dpe.engine._check(dpe.engine.lib().dpe_memcpy_h2d(C.c_void_p(ptr), src.ctypes.data_as(C.c_void_p), C.c_int64(4 * n), None))
    torch.cuda.synchronize()
    bcs.Update(torch.from_numpy(iq).to("cuda:0"), cs)
    bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
    res = bcm.results()
    raw_p = dpe.engine.d2h(bcm.PosScores, W * 1024 * 4, np.float32, None).reshape(W, 1024)
    raw_v = dpe.engine.d2h(bcm.VelScores, W * 800 * 4, np.float32, None).reshape(W, 800)
    ps, vs = bcm.read_scores()
    assert ps.shape == (W, G) and vs.shape == (W, 777)
    assert np.array_equal(raw_p[:, :G], ps) and np.array_equal(raw_v[:, :777], vs)
    assert (raw_p[:, G:] == fill).all() and (raw_v[:, 777:] == fill).all()          # the padding is never written
    assert (ps >= 0).all() and (vs >= 0).all()
    for w in range(W):
        assert res[w]["posIndex"] == int(np.argmax(ps[w])) and res[w]["velIndex"] == int(np.argmax(vs[w]))
    bcm.Stop(); bcs.Stop()


@pytest.mark.parametrize("tpb", [1, 2, 5, 13])
def test_batch_bank_kernel_block_shapes(tpb, monkeypatch):
    """bcs_bank16_kernel sizes its blocks by a cost model (tiles per block) and deals them to the XCDs by
    (window, tile group, SV).  Every shape must give the same banks: 21 windows (21 x nBlk is not a multiple of the 8-block
    dealing unit, so the padding blocks of the launch are exercised) with 1, 2, 5 and all 13 tiles per block (DPE_BCS_TPB16),
    each against the single-window kernel."""
    import torch
    cfg = dpe.workload.CONFIG_R
    W, L, B = 21, 4, 20
    iq, cs, ce, bw = dpe.workload.build_windows(W, cfg["fs"], cfg["S"], cfg["K"], seed=43, amp=cfg["amp"])
    iq_d = torch.from_numpy(iq).to("cuda:0")

    def run(wsel, maxw):
        bcs = dpe.BatchCorrScores(cfg["fs"], samples_per_window=cfg["S"], lag_half_width=L, bin_half_width=B,
                                  max_windows=maxw, max_channels=cfg["K"])
        bcs.Start()
        bcs.Update(iq_d[wsel], cs[wsel])
        code, carr = bcs.read_banks()
        info = bcs.read_info()
        kern = bcs.stage1_kernel
        bcs.Stop()
        return code, carr, info, kern

    monkeypatch.setenv("DPE_BCS_TPB16", str(tpb))
    code, carr, info, kern = run(slice(0, W), W)
    monkeypatch.delenv("DPE_BCS_TPB16")
    assert kern == "bcs_bank16_kernel"
    for w in (0, 10, W - 1):
        c1, f1, i1, k1 = run(slice(w, w + 1), 1)
        assert k1 == "bcs_bank_kernel"
        assert np.abs(code[w] - c1[0]).max() <= 1e-6 * np.abs(c1[0]).max()
        assert np.abs(carr[w] - f1[0]).max() <= 1e-6 * np.abs(f1[0]).max()
        assert np.array_equal(info[0][w], i1[0][0]) and np.array_equal(info[1][w], i1[1][0]) and info[2][w] == i1[2][0]


@pytest.mark.parametrize("pos_out,vel_out,wmean", [(False, True, False), (True, True, True), (False, False, False)])
def test_clamp_variants_of_the_fused_scan(pos_out, vel_out, wmean):
    """The fused scan is instantiated per manifold with or without the range clamp (chosen by a host-side proof that
    every index stays inside the bank).  All combinations against the oracle: scores, arg-max, out-of-window counts."""
    case = helpers.make_case(seed=19, S=12500, K=4, G=3000, amp=200.0, W=2)
    if pos_out:
        case["pos"][:, 3] *= 20.0        # clock offsets of +-2.6 km -> beyond +-4 lags
    if vel_out:
        case["vel"][:, 3] *= 40.0        # clock-drift offsets far beyond +-32 bins
    out = helpers.run_gpu(case, 4, 32, weighted_mean=wmean)
    ref = helpers.run_oracle(case, 4, 32)
    assert (ref["res"][0]["posOutOfWindow"] > 0) == pos_out and (ref["res"][0]["velOutOfWindow"] > 0) == vel_out
    helpers.assert_parity(out, ref, tol=TOL)


def test_fat_finalize_shape_is_bit_identical_to_the_split_shape(monkeypatch):
    """Batches launch bcs_finalize_kernel with one block per (window, SV) doing the code bank and every Doppler-bin
    group; few windows launch 1 + nBinBlk short blocks.  Same arithmetic in the same order: with stage 1 pinned to
    one kernel (DPE_BCS_NO_BANK16, DPE_BCS_NO_FUSE) and one tile partition, a 64-window batch reproduces single-window calls bit for bit."""
    import torch
    monkeypatch.setenv("DPE_BCS_NO_BANK16", "1")
    monkeypatch.setenv("DPE_BCS_NO_FUSE", "1")     # single windows would otherwise take the fused DC-sum form of stage 1
    fs, S, K, W, L, B = 2.5e6, 12500, 8, 64, 4, 24
    iq, cs, ce, bw = dpe.workload.build_windows(W, fs, S, K, seed=53, amp=60.0)
    iq_d = torch.from_numpy(iq).to("cuda:0")

    def run(wsel, maxw):
        bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=maxw, max_channels=K)
        bcs.Start()
        bcs.Update(iq_d[wsel], cs[wsel])
        code, carr = bcs.read_banks()
        info = bcs.read_info()
        bcs.Stop()
        return code, carr, info

    code, carr, info = run(slice(0, W), W)
    for w in (0, 31, W - 1):
        c1, f1, i1 = run(slice(w, w + 1), 1)
        assert np.array_equal(code[w], c1[0]) and np.array_equal(carr[w], f1[0])
        assert np.array_equal(info[1][w], i1[1][0])


@pytest.mark.parametrize("fs,S,L,scale", [(25e6, 125000, 100, 7.0), (2.5e6, 25000, 48, 38.0)])
def test_lag_windows_wider_than_32_samples(fs, S, L, scale):
    """Lag windows beyond +-32 are produced in chunks of 65 lags (the stage-1 kernels re-run against the replica
    delayed by 65 j samples); 25 Msps goes through the boundary-difference kernel, 2.5 Msps through the dense one.
    Banks over the whole window, manifold scores of a grid that really reaches the outer chunks, and the arg-max
    against the oracle."""
    case = helpers.make_case(seed=61, fs=fs, S=S, K=4, G=3000, amp=200.0)
    case["pos"][:, 3] *= scale            # clock-bias offsets that spread the code-lag index over most of +-L
    out = helpers.run_gpu(case, L, 16)
    ref = helpers.run_oracle(case, L, 16)
    assert ref["res"][0]["posOutOfWindow"] == 0
    g = fs * 1.023e6 / (1.023e6 * 299792458.0)
    assert np.abs(case["pos"][:, 3]).max() * g > 40            # the grid reaches beyond the centre chunk
    helpers.assert_parity(out, ref, tol=TOL)


@pytest.mark.parametrize("enable_ekf", [False, True])
def test_closed_loop_with_a_moving_receiver(enable_ekf):
    """System check of the loop the path lives in: a receiver moving at (5, -3, 2) m/s ECEF, samples synthesised from
    the true trajectory, the estimator starting at the handoff position with ZERO velocity.
    * The velocity manifold pulls the velocity state onto the truth within a few windows (to the 2 m/s grid).
    * Position: at 2.5 Msps one sample is 120 m and the reference's score interpolates LINEARLY between integer lags,
      so a sampled correlation triangle peaks at an integer lag for any sub-sample shift -- metre-level motion is
      invisible to the position manifold (a property of the reference algorithm, reproduced by the oracle).  With the
      shipped pass-through the position therefore stays put; with cuEKF's real filter (EnableEKF=true) the velocity
      estimate carries it along through the F matrix."""
    fs, S, K, W = 2.5e6, 50000, 8, 40
    v = np.array([5.0, -3.0, 2.0])
    truth = []
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=71, amp=200.0, velocity=v, truth_out=truth)
    ho = dpe.workload.extend_handoff(dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV), K)
    pos = dpe.synth.uniform_grid(9, 1.0)
    vel = dpe.synth.uniform_grid(9, 2.0)
    fixes, res = dpe.pipeline.run_closed_loop(iq, ho, fs, pos, vel, time_grid=np.unique(pos[:, 3]), K=K, enable_ekf=enable_ekf)
    T = S / fs
    end_truth = np.array(truth)[:, :3] + v * T              # the fix refers to the end of its window
    perr = np.linalg.norm(fixes[:, :3] - end_truth, axis=1)
    verr = np.linalg.norm(fixes[:, 4:7] - v, axis=1)
    assert verr[0] > 5.0 and verr[15:].max() < 2.0, verr    # starts 6.2 m/s off, ends within the grid quantisation
    if enable_ekf:
        assert perr.max() < 2.0, perr                       # carried along by the velocity estimate
    else:
        assert abs(perr[-1] - np.linalg.norm(v) * T * W) < 0.5, perr   # pass-through: the position state does not move
    assert all(r["posOutOfWindow"] == 0 and r["velOutOfWindow"] == 0 for r in res[5:])


def test_parameter_copy_path_for_small_batches():
    """W x K = 2 x 20 = 40 (window, channel) pairs: just above the 37 that travel as kernel arguments, so the
    parameters go through the pinned staging ring + H2D copy with the few-window kernel shapes; repeated Updates
    cycle through all staging slots."""
    case = helpers.make_case(seed=83, S=12500, K=20, G=2048, amp=120.0, W=2)
    ref = helpers.run_oracle(case, 8, 32)
    out = helpers.run_gpu(case, 8, 32, repeats=6)   # more back-to-back calls on one handle than staging slots
    helpers.assert_parity(out, ref, tol=TOL)


def test_repeated_updates_are_reproducible():
    """3000 single-window Updates cycling through 4 windows on one pair of handles (kernel-argument parameters,
    alternating key sets, last-block result publishing, no fences): every result equals the first pass bit for bit.
    (A 150 000-iteration run of the same loop was clean when the publishing scheme was introduced.)"""
    import torch
    cfg = dpe.workload.CONFIG_R
    fs, S, K, L, B, W = cfg["fs"], cfg["S"], cfg["K"], 3, 40, 4
    iq, cs, ce, bw = dpe.workload.build_windows(W, fs, S, K, seed=9, amp=cfg["amp"])
    g = dpe.synth.uniform_grid(9, 1.0)
    iq_d = torch.from_numpy(iq).to("cuda:0")
    bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=K)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, g, g, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=K)
    bcm.Start()

    def one(w):
        bcs.Update(iq_d[w], cs[w])
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[w:w + 1], ce[w])
        r = bcm.results()[0]
        return r["posIndex"], r["velIndex"], r["posScore"], r["velScore"], r["posOutOfWindow"], r["velOutOfWindow"]

    ref = [one(w) for w in range(W)]
    assert all(one(i % W) == ref[i % W] for i in range(3000))
    bcm.Stop(); bcs.Stop()


@pytest.mark.parametrize("W", [1, 2])
def test_large_dc_offset(W):
    """Front ends leave DC offsets far larger than the GPS signal.  (raw - mean) is formed per sample in the batch
    kernels and by moment linearity, M_p[raw w r] - mean M_p[w r], in the single-window kernel -- both must hold the
    usual tolerance with an offset of (+900, -700) LSB on a ~60 LSB signal (W = 1: fused form; W = 2 with K = 20: the
    separate DC-sum kernel)."""
    K = 4 if W == 1 else 20
    case = helpers.make_case(seed=91, S=25000, K=K, G=2048, amp=60.0, W=W)
    for w in case["wins"]:
        iq = w["iq"].astype(np.int32)
        iq[0::2] += 900
        iq[1::2] -= 700
        w["iq"] = np.clip(iq, -32768, 32767).astype(np.int16)
    out = helpers.run_gpu(case, 8, 32)
    ref = helpers.run_oracle(case, 8, 32)
    assert abs(ref["info"][0][0]["mean"].real - 900) < 5 and abs(ref["info"][0][0]["mean"].imag + 700) < 5
    helpers.assert_parity(out, ref, tol=TOL)


def test_hbm_ceiling_diagnostic():
    """dpe_hbm_ceiling: plausible numbers (between 1 and 8 TB/s on an MI355X) and argument errors in the reference's
    0 / -1 convention."""
    cp, tr = dpe.engine.hbm_ceiling(256 << 20, 5)
    assert 1000.0 < cp < 8000.0 and 1000.0 < tr < 8000.0
    with pytest.raises(dpe.DpeError):
        dpe.engine.hbm_ceiling(16, 5)


def test_reference_maximum_grid_size():
    """BCM_MAX_GRID_SIZE = 2 * 75^4 (batchcorrmanifold.h:17): a 75^4 = 31 640 625-point Cartesian position grid plus a
    velocity grid of the same size in one call.  The oracle scores every 997th point (and the neighbourhood of the
    maximum); the arg-max the scan reports must be the first maximum of the scores it wrote."""
    import torch
    dim = 75
    case = helpers.make_case(seed=23, S=12500, K=4, G=16, amp=200.0, grid="uniform")   # grids replaced below
    pos = dpe.synth.uniform_grid(dim, 1.0)
    vel = dpe.synth.uniform_grid(dim, 0.1)
    G = dim ** 4
    assert pos.shape[0] == G
    iq, cs, ce, bw = helpers.pack_gpu_inputs(case)
    L, B = 6, 32
    bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=case["S"], lag_half_width=L, bin_half_width=B, max_channels=4)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(case["fs"], case["S"], bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B,
                                max_channels=4, write_scores=True)
    bcm.Start()
    bcs.Update(torch.from_numpy(iq).to("cuda:0"), cs)
    bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
    r = bcm.results()[0]
    ps, vs = bcm.read_scores()
    bcm.Stop(); bcs.Stop()
    assert r["posOutOfWindow"] == 0 and r["velOutOfWindow"] == 0
    assert r["posIndex"] == int(np.argmax(ps[0])) and r["velIndex"] == int(np.argmax(vs[0]))
    assert r["posScore"] == ps[0].max() and r["velScore"] == vs[0].max()
    # oracle on a strided subset plus the tail and the neighbourhood of the maxima
    for name, grid, sc, best in (("pos", pos, ps[0], r["posIndex"]), ("vel", vel, vs[0], r["velIndex"])):
        idx = np.unique(np.concatenate([np.arange(0, G, 997), np.arange(G - 300, G),
                                        np.clip(np.arange(best - 200, best + 200), 0, G - 1)]))
        sub = dict(case, pos=grid[idx] if name == "pos" else case["pos"], vel=grid[idx] if name == "vel" else case["vel"])
        ref = helpers.run_oracle(sub, L, B)
        want = ref["pos_x" if name == "pos" else "vel"][0]
        assert np.abs(sc[idx] - want).max() < TOL * want.max(), name
        assert sc[best] >= want.max() * (1 - TOL)


@pytest.mark.parametrize("name,W,probe", [("R", 256, (0, 100, 255)), ("H", 128, (0, 5, 127)), ("M", 256, (0, 100, 255))])
def test_bench_configurations_against_the_oracle(oracle, name, W, probe):
    """The exact configurations bench.py times.  R: S = 50000, 8 SVs, two 390625-point rngrid3-format grids, 256 windows per
    call, banks L = 4 / B = 20 (the 16-samples-per-lane bank kernel, the fat finalize shape, the batch scan).  H (--config H):
    25 Msps, S = 500000, 12 SVs, 1e5-point grids, 128 windows per call -- 8 distinct ones repeated with their channel state, as
    bench.py builds them -- L = 31 (bcs_bank_chip2_kernel: its tile length follows from the batch size).  Banks and every 97th
    score of a few windows against the oracle, and for ALL windows the reported arg-max against the first maximum of the
    scores the scan wrote.  M (the extra line of the default bench): config R's windows against the 1e6-point global grids, 256
    windows per call -- the scan's block split and the 2 GB of scores of the benched shape; every 997th score of the probe windows."""
    import torch
    o = oracle
    cfg = {"R": dpe.workload.CONFIG_R, "H": dpe.workload.CONFIG_H, "M": dpe.workload.CONFIG_M}[name]
    fs, S, K, G, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["G"], cfg["L"], cfg["B"]
    distinct = W if name != "H" else 8
    iq, cs, ce, bw = dpe.workload.build_windows(distinct, fs, S, K, seed=0, amp=cfg["amp"])
    if distinct < W:
        iq, cs, ce, bw = (np.concatenate([a] * (W // distinct))[:W] for a in (iq, cs, ce, bw))
    if name == "M":
        pos, vel, _, _, _ = dpe.workload.build_grids_strong(G, 0, 1)      # the global grids bench.py's M line scans on one GPU
    else:
        _, _, pos, vel, _ = dpe.workload.build_grids(G)
    bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=W,
                                max_channels=K, write_scores=True)
    bcm.Start()
    bcs.Update(torch.from_numpy(iq).to("cuda:0"), cs)
    bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
    res = bcm.results()
    code, carr = bcs.read_banks()
    ps, vs = bcm.read_scores()
    kernel = bcs.stage1_kernel
    bcm.Stop(); bcs.Stop()
    assert kernel == ("bcs_bank_chip2_kernel" if name == "H" else "bcs_bank16_kernel")
    C = dpe.engine.carr_fft_len(S)
    for w in range(W):
        assert res[w]["posOutOfWindow"] == 0 and res[w]["velOutOfWindow"] == 0
        assert res[w]["posIndex"] == int(np.argmax(ps[w])) and res[w]["velIndex"] == int(np.argmax(vs[w]))
        assert res[w]["posScore"] == ps[w].max() and res[w]["velScore"] == vs[w].max()
    idx = np.arange(0, G, 997 if name == "M" else 97)
    for w in probe:
        rc, rf = [], []
        for k in range(K):
            c = cs[w, k]
            a, b, _ = o.bcs_sv(iq[w], fs, int(c["prn"]), c["codePhaseStart"], c["carrierPhaseStart"], c["codeFrequency"],
                               c["carrierFrequency"], int(c["cpElapsedStart"]), int(c["cpReference"]), -L, L, -B, B, C)
            rc.append(a); rf.append(b)
        rc, rf = np.stack(rc), np.stack(rf)
        assert np.abs(code[w] - rc).max() < TOL * np.abs(rc).max() and np.abs(carr[w] - rf).max() < TOL * np.abs(rf).max()
        e = ce[w]
        sp, _ = o.bcm_pos(e["satState"], rc, S // 2 - L, bw[w]["xCurrkk1"], pos[idx], bw[w]["enu2ecef"], e["codeFrequency"],
                          e["cpRefTOW"], e["cpElapsedEnd"], e["cpRef"], e["codePhaseEnd"], float(bw[w]["rxTime"]), fs, S, 1,
                          extended=True)
        sv, _ = o.bcm_vel(e["satState"], rf, C // 2 - B, bw[w]["xCurrkk1"], vel[idx], bw[w]["enu2ecef"],
                          e["carrierFrequency"], float(bw[w]["rxTime"]), fs, C, 1, 1)
        assert np.abs(ps[w][idx] - sp).max() < TOL * sp.max() and np.abs(vs[w][idx] - sv).max() < TOL * sv.max()
        # ... and against the FAITHFUL fp64 evaluation (the reference's own rxTime - pr/C rounding, helpers.POS_REF_NOISE)
        spf, _ = o.bcm_pos(e["satState"], rc, S // 2 - L, bw[w]["xCurrkk1"], pos[idx], bw[w]["enu2ecef"], e["codeFrequency"],
                           e["cpRefTOW"], e["cpElapsedEnd"], e["cpRef"], e["codePhaseEnd"], float(bw[w]["rxTime"]), fs, S, 1)
        assert np.abs(ps[w][idx] - spf).max() < helpers.POS_REF_NOISE * spf.max()


def test_config_m_one_gpu_and_a_shard(oracle):
    """BASELINE.json configs[3] on ONE GPU: 2.5 Msps, 8 SVs, the 1e6-point global rngrid3-format grids (rand_grid seed 3 / 4,
    SURVEY 8d) -- every 997th score of both manifolds against the oracle, arg-max == first maximum of the scores written;
    then the [3/8, 4/8) shard of the same grids with its index offset (what rank 3 of 8 scores): identical scores on the
    slice, global indices in the keys."""
    import torch
    o = oracle
    cfg = dpe.workload.CONFIG_M
    fs, S, K, G, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["G"], cfg["L"], cfg["B"]
    W = 4
    iq, cs, ce, bw = dpe.workload.build_windows(W, fs, S, K, seed=0, amp=cfg["amp"])
    pos, vel, _, _, _ = dpe.workload.build_grids_strong(G, 0, 1)
    assert pos.shape == (1000000, 4)
    bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K)
    bcs.Start()
    bcs.Update(torch.from_numpy(iq).to("cuda:0"), cs)
    code, carr = bcs.read_banks()
    C = dpe.engine.carr_fft_len(S)

    def scan(p, v, off):
        bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, p, v, lag_half_width=L, bin_half_width=B, max_windows=W,
                                    max_channels=K, write_scores=True, pos_index_offset=off, vel_index_offset=off)
        bcm.Start()
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
        res = bcm.results()
        ps, vs = bcm.read_scores()
        bcm.Stop()
        return res, ps, vs

    res, ps, vs = scan(pos, vel, 0)
    idx = np.arange(0, G, 997)
    for w in range(W):
        assert res[w]["posOutOfWindow"] == 0 and res[w]["velOutOfWindow"] == 0
        assert res[w]["posIndex"] == int(np.argmax(ps[w])) and res[w]["velIndex"] == int(np.argmax(vs[w]))
        e = ce[w]
        sp, _ = o.bcm_pos(e["satState"], code[w], S // 2 - L, bw[w]["xCurrkk1"], pos[idx], bw[w]["enu2ecef"], e["codeFrequency"],
                          e["cpRefTOW"], e["cpElapsedEnd"], e["cpRef"], e["codePhaseEnd"], float(bw[w]["rxTime"]), fs, S, 1,
                          extended=True)
        sv, _ = o.bcm_vel(e["satState"], carr[w], C // 2 - B, bw[w]["xCurrkk1"], vel[idx], bw[w]["enu2ecef"],
                          e["carrierFrequency"], float(bw[w]["rxTime"]), fs, C, 1, 1)
        assert np.abs(ps[w][idx] - sp).max() < TOL * sp.max() and np.abs(vs[w][idx] - sv).max() < TOL * sv.max()
    b, e_ = dpe.sharding.shard_range(G, 3, 8)
    res3, ps3, vs3 = scan(pos[b:e_], vel[b:e_], b)
    for w in range(W):
        assert np.array_equal(ps3[w], ps[w][b:e_]) and np.array_equal(vs3[w], vs[w][b:e_])
        assert res3[w]["posIndex"] == b + int(np.argmax(ps[w][b:e_])) and res3[w]["velIndex"] == b + int(np.argmax(vs[w][b:e_]))
    bcs.Stop()


@pytest.mark.parametrize("S,W", [(8192, 3), (16384, 6)])
def test_reference_pair_mode_reproduces_the_double_counting_branch(S, W):
    """dpe_bcm_config.referencePair: with S / 2 a power of two the reference's floor(idx) / floor(idx + 1) pair
    (batchcorrmanifold.cu:1798-1812) double-counts at the grid points whose first-channel index collapses onto the value one
    fp64 step below 2^m (the zero-offset centre of a closed loop and its neighbours inside the reference's 1.4e-4-sample
    index noise).  In this mode the scores there equal the FAITHFUL oracle's (no point set aside) and the arg-max is the
    faithful one; by default the continuous interpolation is evaluated and those points differ by ~3 %."""
    case = helpers.make_case(seed=1, S=S, K=6, G=625, amp=200.0, W=W, grid="uniform")
    ref = helpers.run_oracle(case, 4, 12)
    quirky = [w for w in range(W) if len(ref["pos_quirk"][w])]
    assert quirky, "the case must exercise the branch"
    out = helpers.run_gpu(case, 4, 12, reference_pair=True)
    plain = helpers.run_gpu(case, 4, 12)
    for w in range(W):
        r, g = ref["pos"][w], out["pos"][w]
        assert np.abs(g - r).max() < helpers.POS_REF_NOISE * r.max()           # every point, none excluded
        q = ref["pos_quirk"][w]
        if len(q):
            assert np.abs(g[q] - r[q]).max() < 5e-6 * r.max()                   # patched points: the reference's own expression
            assert np.abs(plain["pos"][w][q] - r[q]).max() > 1e-3 * r.max()     # ... which the default mode does not follow
        assert out["res"][w]["posIndex"] == ref["res"][w]["posIndex"]
        assert out["res"][w]["posScore"] == g.max()
        assert np.abs(out["res"][w]["zVal"] - ref["res"][w]["zVal"]).max() < 1e-6
        assert np.abs(out["vel"][w] - ref["vel"][w]).max() < TOL * ref["vel"][w].max()
    # S / 2 not a power of two: the mode changes nothing
    case2 = helpers.make_case(seed=2, S=12500, K=4, G=625, amp=200.0, W=2, grid="uniform")
    a, b = helpers.run_gpu(case2, 4, 20, reference_pair=True), helpers.run_gpu(case2, 4, 20)
    for w in range(2):
        assert np.array_equal(a["pos"][w], b["pos"][w]) and a["res"][w]["posIndex"] == b["res"][w]["posIndex"]


def test_reference_pair_mode_with_the_device_ports(oracle):
    """referencePair through dpe_bcm_update_dev: candidates, the reference's fp64 expression (batchcorrmanifold.cu:1760-1816) from the
    port arrays, the patch and the re-derived arg-max all on the device, nothing read back -- the scores of the host form of the
    mode (which the test above holds against the faithful oracle), the same arg-max and fix; S / 2 = 4096."""
    import torch
    S, K, L, B = 8192, 6, 4, 12
    case = helpers.make_case(seed=1, S=S, K=K, G=625, amp=200.0, W=3, grid="uniform")
    ref = helpers.run_oracle(case, L, B)
    fs = case["fs"]
    iq, cs, ce, bw = helpers.pack_gpu_inputs(case)
    dev = torch.device("cuda:0")
    tg = np.zeros(1)
    checked = 0
    for w in range(3):
        q = ref["pos_quirk"][w]
        win = case["wins"][w]
        out = {}
        for form in ("host", "dev"):
            bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=K)
            bcs.Start()
            bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, case["pos"], case["vel"], lag_half_width=L, bin_half_width=B, max_channels=K,
                                        weighted_mean=False, reference_pair=True)
            bcm.Start()
            bcs.Update(torch.from_numpy(iq[w]).to(dev), cs[w:w + 1])
            if form == "host":
                bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[w:w + 1], ce[w:w + 1])
            else:
                def d(a, dt):
                    return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dt)).to(dev)
                keep = dict(x=d(win["centre"], np.float64), R=d(np.asarray(win["R"]).ravel(), np.float64), sat=d(win["sat"][:, None, :], np.float64),
                            rcE=d(win["rcEnd"], np.float64), fc=d(win["fc"], np.float64), fi=d(win["fi"], np.float64),
                            tow=d(win["cpRefTOW"], np.int32), elaE=d(win["cpElaEnd"], np.int32), ref=d(win["cpRef"], np.int32), ds=d([1], np.int32))
                bcm.UpdateDev(bcs.CodeScores, bcs.CarrScores, K,
                              dict(xCurrkk1=keep["x"], enu2ecef=keep["R"], satStates=keep["sat"], codePhaseEnd=keep["rcE"], codeFrequency=keep["fc"],
                                   carrierFrequency=keep["fi"], cpRefTOW=keep["tow"], cpElapsedEnd=keep["elaE"], cpRef=keep["ref"], dopplerSign=keep["ds"]),
                              1, float(win["rxTime"]))
            r = bcm.results()[0]
            ps, _ = bcm.read_scores()
            out[form] = (r, ps[0].copy())
            bcm.Stop(); bcs.Stop()
        (rh, ph), (rd, pd) = out["host"], out["dev"]
        assert np.abs(pd - ph).max() < 2e-6 * ph.max()
        assert rd["posIndex"] == rh["posIndex"] == ref["res"][w]["posIndex"] and rd["posScore"] == pd.max()
        assert np.abs(rd["zVal"] - rh["zVal"]).max() < 1e-9
        if len(q):
            assert np.abs(pd[q] - ref["pos"][w][q]).max() < 5e-6 * ref["pos"][w].max()     # the patched points follow the reference's expression
            checked += 1
    assert checked, "the case must exercise the branch"
