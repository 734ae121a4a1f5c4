import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
cfg = dpe.workload.CONFIG_R
W = 8
iq, cs, ce, bw = dpe.workload.build_windows(W, cfg["fs"], cfg["S"], cfg["K"], seed=23, amp=cfg["amp"])
L, B = cfg["L"], cfg["B"]
iq_d = torch.from_numpy(iq).to("cuda:0")
def run(graph, stream, nring, same_params=False, same_samples=False):
    ring = [torch.empty_like(iq_d[0]) for _ in range(nring)]
    bcs = dpe.BatchCorrScores(cfg["fs"], samples_per_window=cfg["S"], lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=cfg["K"])
    bcs.Start()
    bcs.set_graph(graph)
    out = []
    for w in range(W):
        ws = 0 if same_samples else w
        wp = 0 if same_params else w
        slot = ring[w % nring]
        slot.copy_(iq_d[ws]); torch.cuda.synchronize()
        bcs.Update(slot, cs[wp], stream=stream)
        code, carr = bcs.read_banks(stream=stream)
        out.append((code.copy(), carr.copy()))
    bcs.Stop()
    return out
st = dpe.engine.Stream()
for nring in (1, 2, 3):
    for sp, ss in ((False, False), (True, False), (False, True), (True, True)):
        e = run(False, st, nring, sp, ss)
        v = run(True, st, nring, sp, ss)
        print("ring", nring, "same_params", sp, "same_samples", ss, [(bool(np.array_equal(a[0], b[0])), bool(np.array_equal(a[1], b[1]))) for a, b in zip(e, v)])
e = run(False, st, 2); v = run(True, st, 2)
print(e[3][1][0, 0, :6]); print(v[3][1][0, 0, :6])
