import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
cfg = dpe.workload.CONFIG_R
W = 6
iq, cs, ce, bw = dpe.workload.build_windows(W, cfg["fs"], cfg["S"], cfg["K"], seed=23, amp=cfg["amp"])
_, _, pos, vel, _ = dpe.workload.build_grids(6561)
L, B = cfg["L"], cfg["B"]
iq_d = torch.from_numpy(iq).to("cuda:0")
ring = [torch.empty_like(iq_d[0]) for _ in range(2)]
def run(graph, stream, gb=True, gm=True):
    bcs = dpe.BatchCorrScores(cfg["fs"], samples_per_window=cfg["S"], lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=cfg["K"])
    bcs.Start()
    bcm = dpe.BatchCorrManifold(cfg["fs"], cfg["S"], bcs.NumFFTPoints, pos, vel, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=cfg["K"])
    bcm.Start()
    bcs.set_graph(graph and gb); bcm.set_graph(graph and gm)
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
e = run(False, st)
for name, v in (("both", run(True, st)), ("bcs only", run(True, st, True, False)), ("bcm only", run(True, st, False, True)), ("null", run(True, None))):
    print(name)
    for w in range(W):
        print("  w", w, [bool(np.array_equal(x, y)) for x, y in zip(e[w][:4], v[w][:4])], e[w][4]["posIndex"], v[w][4]["posIndex"],
              float(np.abs(e[w][0]-v[w][0]).max()), float(np.abs(e[w][2]-v[w][2]).max()))
v = run(True, st, True, False)
for w in range(W):
    d = np.abs(e[w][1] - v[w][1])
    print("w", w, "carr shape", d.shape, "max", d.max(), "rel", d.max()/np.abs(e[w][1]).max(), "per-chan", d.reshape(-1, d.shape[-1]).max(axis=1)[:8])
    print("   cs", cs[w]["cpElapsedStart"][:3] if "cpElapsedStart" in cs.dtype.names else cs.dtype.names)
