import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
cfg = dpe.workload.CONFIG_R
W = 5
iq, cs, ce, bw = dpe.workload.build_windows(W, cfg["fs"], cfg["S"], cfg["K"], seed=23, amp=cfg["amp"])
L, B = cfg["L"], cfg["B"]
iq_d = torch.from_numpy(iq).to("cuda:0")
def run(graph, stream):
    bcs = dpe.BatchCorrScores(cfg["fs"], samples_per_window=cfg["S"], lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=cfg["K"])
    bcs.Start()
    bcs.set_graph(graph)
    out = []
    buf = torch.empty_like(iq_d[0])
    for w in range(W):
        buf.copy_(iq_d[w]); torch.cuda.synchronize()
        bcs.Update(buf, cs[w], stream=stream)
        code, carr = bcs.read_banks(stream=stream)
        out.append((code.copy(), carr.copy()))
    bcs.Stop()
    return out
st = dpe.engine.Stream()
e = run(False, st); v = run(True, st)
print(os.environ.get("DPE_DBG_CAPTURE_FROM"), [(bool(np.array_equal(a[0], b[0])), bool(np.array_equal(a[1], b[1]))) for a, b in zip(e, v)])
