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
    for w in range(W):
        bcs.Update(iq_d[0], cs[0], stream=stream)
        idx, nfl, mean = bcs.read_info(stream=stream)
        print(graph, w, mean, idx[0][:3], nfl[0][:3])
    bcs.Stop()
st = dpe.engine.Stream()
run(False, st); run(True, st)
