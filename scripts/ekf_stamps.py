#!/usr/bin/env python3
"""Where chm_k1 spends its time with the filter on: runs the device-resident loop (25^4 grids, EnableEKF = true) with a library built
with -DDPE_EXPERIMENTS -DDPE_EKF_STAMPS (scripts/build_variant.sh stamps "-DDPE_EXPERIMENTS -DDPE_EKF_STAMPS" dpe_chanmgr), whose
measurement kernel leaves the shader-clock lengths of its phases in the fix record's two out-of-window counts.
    DPE_LIB_PATH=scratch/ab/stamps/libdpe_hip.so python scripts/ekf_stamps.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import navlab_dpe_sdr_amd as dpe  # noqa: E402
from navlab_dpe_sdr_amd import engine, pipeline  # noqa: E402

NAMES = ["0-1 measurement (keys, grid rows)", "1-2 S, T, y", "2-3 factorisation", "3-4 substitutions", "4-5 K = T Sinv, x, I - K",
         "5-6 (I - K) P", "6-7 low-pass / sqrt", "7-8 Q, x+, P+"]


def main():
    import torch
    W, fs, S, K = 200, 2.5e6, 50000, 8
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0)
    ho = dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV)
    _, _, pos, vel, _ = dpe.workload.build_grids(390625)
    nfft = engine.carr_fft_len(S)
    L, B = pipeline.bank_half_widths(pos, vel, fs, nfft)
    bcs = engine.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcs.Start()
    bcm = engine.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, LPower=1, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcm.Start()
    cm = engine.ChanMgrDev.from_handoff(ho, S / fs, K, (0.0,))
    cm.attach(bcs, bcm, 256)
    x = np.array(ho["X_ECEF"], dtype=np.float64).copy()
    cm.set_ekf(S / fs, x)
    iq_d = torch.from_numpy(np.ascontiguousarray(iq)).to("cuda:0")
    cm.Start(x, None)
    for w in range(W):
        bcs.UpdatePrepared(iq_d[w], K, None)
        bcm.UpdatePrepared(bcs.CodeScores, bcs.CarrScores, K, None)
        cm.step(None)
    recs = [cm.fix(w) for w in range(W)]
    ph = np.zeros((W, 8))
    for w, r in enumerate(recs):
        lo, hi = int(r["posOutOfWindow"]) & (2 ** 64 - 1), int(r["velOutOfWindow"]) & (2 ** 64 - 1)
        for i in range(4):
            ph[w, i] = 4 * ((lo >> (16 * i)) & 0xFFFF)
            ph[w, 4 + i] = 4 * ((hi >> (16 * i)) & 0xFFFF)
    med = np.median(ph[W // 2:], axis=0)
    for n, m in zip(NAMES, med):
        print("%-36s %7.0f clocks" % (n, m))
    print("sum %.0f clocks (s_memtime counts the constant 100 MHz reference on this part if the numbers look like tens, not thousands)" % med.sum())
    cm.Stop(); bcm.Stop(); bcs.Stop()


if __name__ == "__main__":
    main()
