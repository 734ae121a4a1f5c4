"""Stress of dpe_pipe (batches in flight): N random batches -- window count and contents changing from batch to batch -- dealt to three
lanes back to back, every batch collected only when its lane is about to be dealt again; each result set must equal the one-stream
path's for that batch bit for bit.   usage: python scripts/stress_pipe.py [N=3000]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import navlab_dpe_sdr_amd as dpe

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfg = dpe.workload.CONFIG_R
fs, S, K, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["L"], cfg["B"]
Wmax, G = 24, 20000
iq, cs, ce, bw = dpe.workload.build_windows(Wmax, fs, S, K, seed=31, amp=cfg["amp"])
_, _, pos, vel, _ = dpe.workload.build_grids(G)
iq_d = torch.from_numpy(iq).to("cuda:0")
rng = np.random.Generator(np.random.PCG64(7))
# the batches: (first window, count); reference results from a lone handle pair on the default stream
shapes = [(int(a), int(n)) for a, n in zip(rng.integers(0, Wmax - 1, 40), rng.integers(1, Wmax + 1, 40))]
shapes = [(a, min(n, Wmax - a)) for a, n in shapes]
bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=Wmax, max_channels=K)
bcs.Start()
bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=Wmax, max_channels=K)
bcm.Start()
ref = []
for a, n in shapes:
    bcs.Update(iq_d[a:a + n], cs[a:a + n])
    bcm.Update(bcs.CodeScores, bcs.CarrScores, bw[a:a + n], ce[a:a + n])
    ref.append([(r["posIndex"], r["velIndex"], r["posScore"], r["velScore"], tuple(r["zVal"])) for r in bcm.results()])
bcm.Stop(); bcs.Stop()
pipe = dpe.Pipe(fs, S, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=Wmax, max_channels=K, in_flight=3)
st = torch.cuda.current_stream()
pending, bad, t0 = [], 0, time.time()
for i in range(N):
    if len(pending) == 3:
        t, j = pending.pop(0)
        got = [(r["posIndex"], r["velIndex"], r["posScore"], r["velScore"], tuple(r["zVal"])) for r in pipe.results(t)]
        bad += got != ref[j]
    j = int(rng.integers(0, len(shapes)))
    a, n = shapes[j]
    pending.append((pipe.submit(iq_d[a:a + n], cs[a:a + n], bw[a:a + n], ce[a:a + n], stream=st), j))
for t, j in pending:
    got = [(r["posIndex"], r["velIndex"], r["posScore"], r["velScore"], tuple(r["zVal"])) for r in pipe.results(t)]
    bad += got != ref[j]
pipe.close()
print("stress_pipe: %d batches of 1..%d windows over 3 lanes, %d mismatching result sets, %.1f s" % (N, Wmax, bad, time.time() - t0))
sys.exit(1 if bad else 0)
