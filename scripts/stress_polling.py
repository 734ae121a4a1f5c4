#!/usr/bin/env python3
"""Soak test of the result hand-over: N single-window Update/results round trips (the fix is polled from the pinned
mirror, alternating key sets) over four windows with known answers, then N/50 queued 64-window batches.  Any stale or
torn result shows up as a mismatch.      python scripts/stress_polling.py [N=100000]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
cfg = dpe.workload.CONFIG_R
fs, S, K, L, B = cfg["fs"], cfg["S"], cfg["K"], 3, 40
W = 4
iq, cs, ce, bw = dpe.workload.build_windows(W, fs, S, K, seed=9, amp=cfg["amp"])
pos = dpe.synth.uniform_grid(25, 1.0); vel = dpe.synth.uniform_grid(25, 1.0)
iq_d = torch.from_numpy(iq).to("cuda:0")
st = dpe.engine.Stream()
bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=K); bcs.Start()
bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=K, write_scores=True); bcm.Start()
ref = []
for w in range(W):
    bcs.Update(iq_d[w], cs[w], stream=st); bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[w:w+1], ce[w], stream=st)
    r = bcm.results(stream=st)[0]; ref.append((r["posIndex"], r["velIndex"], r["posScore"], r["velScore"]))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bad = 0
t0 = time.perf_counter()
for i in range(N):
    w = i % W
    bcs.Update(iq_d[w], cs[w], stream=st); bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[w:w+1], ce[w], stream=st)
    r = bcm.results(stream=st)[0]
    if (r["posIndex"], r["velIndex"], r["posScore"], r["velScore"]) != ref[w]:
        bad += 1
        if bad < 5: print("MISMATCH at", i, r["posIndex"], r["velIndex"], ref[w])
print("W=1 closed-loop style:", N, "iterations,", bad, "mismatches,", (time.perf_counter()-t0)/N*1e6, "us/iter")
bcm.Stop(); bcs.Stop()
# batch mode
Wb = 64
iqb, csb, ceb, bwb = dpe.workload.build_windows(Wb, fs, S, K, seed=10, amp=cfg["amp"])
_, _, posr, velr, _ = dpe.workload.build_grids(100000)
iqb_d = torch.from_numpy(iqb).to("cuda:0")
bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=4, bin_half_width=20, max_windows=Wb, max_channels=K); bcs.Start()
bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, posr, velr, lag_half_width=4, bin_half_width=20, max_windows=Wb, max_channels=K, write_scores=True); bcm.Start()
bcs.Update(iqb_d, csb, stream=st); bcm.Update(bcs.CodeScores, bcs.CarrScores, bwb, ceb, stream=st)
ref = [(r["posIndex"], r["velIndex"], r["posScore"], r["velScore"]) for r in bcm.results(stream=st)]
bad = 0
M = N // 50
for i in range(M):
    bcs.Update(iqb_d, csb, stream=st); bcm.Update(bcs.CodeScores, bcs.CarrScores, bwb, ceb, stream=st)
    if i % 7 == 0 or i == M - 1:     # results() synchronises: mostly let several steps queue up
        got = [(r["posIndex"], r["velIndex"], r["posScore"], r["velScore"]) for r in bcm.results(stream=st)]
        if got != ref:
            bad += 1
print("batch W=64:", M, "steps,", bad, "mismatching result sets")
