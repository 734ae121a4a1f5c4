#!/usr/bin/env python3
"""Monte-Carlo of the closed loop over random initial receiver offsets -- the reference's own manual check
(cudarecv/src/main.cu:105-223, "Automated method 2": runs with the initial position shifted by 50..80 m in a random
horizontal direction).  Static synthetic recording whose truth is the handoff position, 25^4 x 1 m grids as in
dpeflow.cpp:83-85, fix fed back through cuChanMgr.  Two estimators per run:

  ML    the grid point with the highest score -- what BatchCorrManifold::Update has active (batchcorrmanifold.cu:2565-2596)
  mean  the score-weighted mean of the grid -- its commented-out "Method 1" (:2546-2563), PyGNSS' default (receiver.py:317-318)

What to expect (and what this prints): the ML fix does not move.  The manifold interpolates the sample-rate correlation
LINEARLY, a piecewise-linear function has its maxima at its nodes, and the grid centre sits on a node by construction
(cuChanMgr back-calculates the code phase from the very state the grid is centred on), so within +-12 m = +-0.1 sample of
it the centre always scores highest; only offsets beyond half a sample (60 m at 2.5 Msps) or a wider grid (PyGNSS' +-110 m
spread grid) reach another node.  The weighted mean sees the asymmetry of the two slopes and drifts towards the truth
(~1.6 m/s with LPower 1).  Same arithmetic as the reference (parity tests); this script documents the behaviour.

    python scripts/monte_carlo.py [runs=8] [windows=60]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import navlab_dpe_sdr_amd as dpe  # noqa: E402
from navlab_dpe_sdr_amd import engine  # noqa: E402


def main():
    import torch
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    fs, S, K, L, B = 2.5e6, 50000, 8, 6, 40
    iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0)
    ho = dpe.handoff.read_handoff(dpe.workload.HANDOFF_CSV)
    pos = dpe.synth.uniform_grid(25, 1.0)
    vel = dpe.synth.uniform_grid(25, 1.0)
    X = ho["X_ECEF"].copy()
    up = X[:3] / np.linalg.norm(X[:3])
    east = np.cross([0.0, 0.0, 1.0], up)
    east /= np.linalg.norm(east)
    north = np.cross(up, east)
    iq_d = torch.from_numpy(iq).to("cuda:0")
    bcs = engine.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_channels=K)
    bcs.Start()
    bcm = engine.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_channels=K,
                                   weighted_mean=True)
    bcm.Start()
    rng = np.random.Generator(np.random.PCG64(2024))
    for r in range(runs):
        mag = rng.uniform(50.0, 80.0) * rng.choice([-1.0, 1.0])          # main.cu: |shift| in [50, 80] m
        th = rng.uniform(0.0, 2.0 * np.pi)
        x0 = X.copy()
        x0[:3] += mag * (np.sin(th) * north + np.cos(th) * east)
        line = "run %2d: |shift| %5.1f m" % (r, abs(mag))
        for est in ("ML", "mean"):
            cm = engine.ChanMgr.from_handoff(ho, S / fs, K)
            xk = x0.copy()
            for w in range(W):
                (cm.Start if w == 0 else cm.Update)(xk, xk, np.zeros(1))
                cs, ce, bw = cm.outputs()
                bcs.Update(iq_d[w], cs)
                bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
                res = bcm.results()[0]
                xk = (res["zVal"] if est == "ML" else res["zValMean"]).copy()
            cm.Stop()
            line += "   %s: error after %d windows %.2f m" % (est, W, np.linalg.norm(xk[:3] - X[:3]))
        print(line)
    bcm.Stop()
    bcs.Stop()


if __name__ == "__main__":
    main()
