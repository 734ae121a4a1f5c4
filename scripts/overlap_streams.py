"""Experiment (usage: overlap_streams.py [R|H]): config R, 256 windows per step as ONE chain (bench.py's form) against TWO half-batches on two streams
(each its own handle pair), so that stage 1 of one half runs beside the scan of the other."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import navlab_dpe_sdr_amd as dpe

name = sys.argv[1] if len(sys.argv) > 1 else "R"
cfg = dict({"R": dpe.workload.CONFIG_R, "H": dpe.workload.CONFIG_H}[name])
fs, S, K, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["L"], cfg["B"]
W = 256 if name == "R" else 128
dev = torch.device("cuda:0")
distinct = W if name == "R" else 8          # (H: 8 distinct windows repeated, as in bench.py)
iq, cs, ce, bw = dpe.workload.build_windows(distinct, fs, S, K, seed=0, amp=cfg["amp"])
if distinct < W:
    iq, cs, ce, bw = (np.concatenate([a] * (W // distinct)) for a in (iq, cs, ce, bw))
pos_g, vel_g, pos, vel, off = dpe.workload.build_grids(cfg["G"], 0, 1)
iq_d = torch.from_numpy(np.ascontiguousarray(iq)).to(dev)

def make(Wl):
    bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=Wl, max_channels=K)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(fs, S, bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=Wl,
                                max_channels=K, write_scores=True)
    bcm.Start()
    return bcs, bcm

def timeit(step, n=60, reps=5):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 1.5
    while time.perf_counter() < t_end:
        step()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    return float(np.median(out)), out

res = {}
# one chain
bcs, bcm = make(W)
s0 = torch.cuda.current_stream()
def step1():
    bcs.Update(iq_d, cs, stream=s0)
    bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce, stream=s0)
res["one chain of %d" % W] = timeit(step1)
r_full = bcm.results()
bcm.Stop(); bcs.Stop()

for parts in (2, 4):
    Wp = W // parts
    hs = [make(Wp) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    sl = [slice(i * Wp, (i + 1) * Wp) for i in range(parts)]
    iqs = [iq_d[s] for s in sl]
    css = [np.ascontiguousarray(cs[s]) for s in sl]
    ces = [np.ascontiguousarray(ce[s]) for s in sl]
    bws = [np.ascontiguousarray(bw[s]) for s in sl]
    def stepn():
        for i in range(parts):
            b, m = hs[i]
            b.Update(iqs[i], css[i], stream=streams[i])
            m.Update(b.CodeScores, b.CarrScores, bws[i], ces[i], stream=streams[i])
    res["%d chains of %d on %d streams" % (parts, Wp, parts)] = timeit(stepn)
    got = []
    for b, m in hs:
        got += m.results()
    assert all(a["posIndex"] == b_["posIndex"] and a["velIndex"] == b_["velIndex"] for a, b_ in zip(got, r_full))
    # the same split on ONE stream (no overlap): what splitting alone costs
    def steps1():
        for i in range(parts):
            b, m = hs[i]
            b.Update(iqs[i], css[i], stream=s0)
            m.Update(b.CodeScores, b.CarrScores, bws[i], ces[i], stream=s0)
    res["%d chains of %d on one stream" % (parts, Wp)] = timeit(steps1)
    for b, m in hs:
        m.Stop(); b.Stop()
# two FULL chains, alternate steps on alternate streams: stage 1 of step n + 1 beside the scan of step n, nothing split
for depth in (2, 3):
    hs = [make(W) for _ in range(depth)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    cnt = [0]
    def stepalt():
        i = cnt[0] % depth
        cnt[0] += 1
        b, m = hs[i]
        b.Update(iq_d, cs, stream=streams[i])
        m.Update(b.CodeScores, b.CarrScores, bw, ce, stream=streams[i])
    res["%d chains of %d, steps alternate" % (depth, W)] = timeit(stepalt)
    got = hs[0][1].results()
    assert all(a["posIndex"] == b_["posIndex"] and a["velIndex"] == b_["velIndex"] for a, b_ in zip(got, r_full))
    for b, m in hs:
        m.Stop(); b.Stop()
for k, v in res.items():
    print("%-36s %.4f ms per %d windows   %s" % (k, v[0], W, ["%.4f" % x for x in v[1]]))
