"""Closed-loop DPE iteration on top of the C-ABI -- the Python twin of host/dpe_flow_main.cpp and of
Flow::FlowThread's module order (cudarecv/dsp/src/flow.cu:122-137; dpeflow.cpp:55-62):
SampleBlock -> BatchCorrScores -> BatchCorrManifold -> cuEKF(pass-through) -> cuChanMgr."""
import numpy as np

from . import engine


def bank_half_widths(pos_grid, vel_grid, fs, nfft):
    """INTEGRATION.md section 3 (same rule as host/grids.hpp::bank_half_widths)."""
    ep = (np.linalg.norm(pos_grid[:, :3], axis=1) + np.abs(pos_grid[:, 3])).max()
    ev = (np.linalg.norm(vel_grid[:, :3], axis=1) + np.abs(vel_grid[:, 3])).max()
    L = int(np.ceil(ep * fs / 299792458.0)) + 2
    B = int(np.ceil(ev * (nfft / fs) * 1.57542e9 / 299792458.0)) + 3
    return L, B


def run_closed_loop(iq_windows, ho, fs, pos_grid, vel_grid, time_grid=(0.0,), init_delta=(0, 0, 0, 0), K=None,
                    lpower=1, enable_ekf=False, reference_pair=False, keep_scores=False, couple_velocity=True):
    """iq_windows: int16 [W, 2S] (host).  Returns fixes [W, 8] (= xCurrk1k1 per window) and the raw
    per-window result dicts.  One window per Update, fix fed back to the channel manager.
    enable_ekf: route the fix through cuEKF's real filter (EnableEKF=true) instead of the shipped pass-through.
    reference_pair: dpe_bcm_config.referencePair; keep_scores: every result dict also carries the window's position scores;
    couple_velocity: the filter's F couples position and velocity over one window (cuekf.cu:111-143) or is the identity (ekf.py:47)."""
    import torch
    iq_windows = np.ascontiguousarray(iq_windows)
    W, S2 = iq_windows.shape
    S = S2 // 2
    K = len(ho["prn_list"]) if K is None else K
    nfft = engine.carr_fft_len(S)
    L, B = bank_half_widths(pos_grid, vel_grid, fs, nfft)
    bcs = engine.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcs.Start()
    bcm = engine.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos_grid, vel_grid, LPower=lpower, lag_half_width=L,
                                   bin_half_width=B, max_channels=K, reference_pair=reference_pair)
    bcm.Start()
    cm = engine.ChanMgr.from_handoff(ho, S / fs, K)
    x = np.array(ho["X_ECEF"], dtype=np.float64).copy()
    x[:4] += np.asarray(init_delta, dtype=np.float64)
    iq_d = torch.from_numpy(iq_windows).to("cuda:0")
    fixes, results = np.zeros((W, 8)), []
    ekf = engine.cuEKF(x, SampleLength=S / fs, EnableEKF=enable_ekf, couple_velocity=couple_velocity)
    xk1k1, xkk1 = x, x
    for w in range(W):
        (cm.Start if w == 0 else cm.Update)(xk1k1, xkk1, time_grid)
        cs, ce, bw = cm.outputs()
        bcs.Update(iq_d[w], cs)
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
        r = bcm.results()[0]
        if keep_scores:
            r["posScores"] = bcm.read_scores()[0][0].copy()
        ekf.Update(r["zVal"], r["RVal"])    # EKF_PassMeas (the ML point is the new state) or the filter
        xk1k1, xkk1 = ekf.xCurrk1k1.copy(), ekf.xCurrkk1.copy()
        fixes[w] = xk1k1
        results.append(r)
    ekf.Stop()
    cm.Stop()
    bcm.Stop()
    bcs.Stop()
    return fixes, results


def run_device_loop(iq_windows, ho, fs, pos_grid, vel_grid, time_grid=(0.0,), init_delta=(0, 0, 0, 0), K=None, lpower=1,
                    ring_depth=64, stream=None, reference_pair=False, keep_scores=False, enable_ekf=False, couple_velocity=True):
    """The same loop with nothing read back per window: the channel manager lives on the device (engine.ChanMgrDev), forms
    the measurement from the scan's keys, passes it through and writes the next window's parameter blocks; the host enqueues
        BatchCorrScores.UpdatePrepared -> BatchCorrManifold.UpdatePrepared -> ChanMgrDev.step
    for every window and collects the fixes from the pinned ring afterwards (at most ring_depth - 1 windows ahead).
    reference_pair: dpe_bcm_config.referencePair (the prepared form re-evaluates from the attached manager's port arrays);
    keep_scores (tests): waits for every window and keeps its position scores, its code banks and the channel manager's outputs the
    window was scored with (`inputs` = ChanMgrDev.outputs() before the window) -- the loop then does read back.
    enable_ekf: cuEKF's filter inside the measurement kernel (dpe_chm_dev_set_ekf) instead of the pass-through; the fixes are then x_k|k."""
    import torch
    iq_windows = np.ascontiguousarray(iq_windows)
    W, S2 = iq_windows.shape
    S = S2 // 2
    K = len(ho["prn_list"]) if K is None else K
    nfft = engine.carr_fft_len(S)
    L, B = bank_half_widths(pos_grid, vel_grid, fs, nfft)
    bcs = engine.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcs.Start()
    bcm = engine.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos_grid, vel_grid, LPower=lpower, lag_half_width=L,
                                   bin_half_width=B, max_channels=K, reference_pair=reference_pair)
    bcm.Start()
    cm = engine.ChanMgrDev.from_handoff(ho, S / fs, K, time_grid)
    cm.attach(bcs, bcm, ring_depth)
    x = np.array(ho["X_ECEF"], dtype=np.float64).copy()
    x[:4] += np.asarray(init_delta, dtype=np.float64)
    iq_d = torch.from_numpy(iq_windows).to("cuda:0")
    if enable_ekf:
        cm.set_ekf(S / fs, x, couple_velocity=couple_velocity)
    cm.Start(x, stream)
    fixes, results, got = np.zeros((W, 8)), [], 0
    for w in range(W):
        while w - got >= ring_depth - 1:          # never more than the ring holds ahead of the fixes already collected
            results.append(cm.fix(got))
            got += 1
        inputs = cm.outputs(stream=stream) if keep_scores else None     # (the previous window's time update has run: fix() flushed it)
        bcs.UpdatePrepared(iq_d[w], K, stream)
        bcm.UpdatePrepared(bcs.CodeScores, bcs.CarrScores, K, stream)
        cm.step(stream)
        if keep_scores:
            while got <= w:
                results.append(cm.fix(got))
                got += 1
            results[w]["posScores"] = bcm.read_scores(stream)[0][0].copy()
            results[w]["codeBank"] = bcs.read_banks(stream)[0][0].copy()
            results[w]["inputs"] = inputs
    while got < W:
        results.append(cm.fix(got))
        got += 1
    for w, r in enumerate(results):
        fixes[w] = r["zVal"]
    cm.outputs()
    status = cm.status
    cm.Stop()
    bcm.Stop()
    bcs.Stop()
    return fixes, results, status
