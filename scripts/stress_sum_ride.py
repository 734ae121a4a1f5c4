"""Stress of the DC sums riding in the chip2 launch: thousands of launches with changing batch sizes; every launch's means must equal
the first launch's for the same windows, device status 0."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import navlab_dpe_sdr_amd as dpe
cfg = dpe.workload.CONFIG_H
fs, S, K, L, B = cfg["fs"], cfg["S"], cfg["K"], cfg["L"], cfg["B"]
W = 128
iq, cs, ce, bw = dpe.workload.build_windows(8, fs, S, K, seed=0, amp=cfg["amp"])
iq, cs = (np.concatenate([a] * (W // 8)) for a in (iq, cs))
d = torch.from_numpy(iq).to("cuda:0")
bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K)
bcs.Start()
bcs.Update(d, cs)
ref_mean = bcs.read_info()[2].copy()
ref_code, ref_carr = (a.copy() for a in bcs.read_banks())
rng = np.random.default_rng(1)
refs = {}
t0 = time.time()
n = 0
while time.time() - t0 < float(os.environ.get("STRESS_S", "120")):
    nw = int(rng.choice([48, 64, 96, 128, 128, 128]))
    for _ in range(int(rng.integers(1, 40))):
        bcs.Update(d[:nw], cs[:nw])
        n += 1
    m = bcs.read_info()[2][:nw]
    assert np.array_equal(m, ref_mean[:nw]), "means differ after %d launches" % n
    code, carr = bcs.read_banks()
    if nw not in refs:   # (the tile length follows from the batch size: banks are bit-identical per batch size, equal to rounding across)
        refs[nw] = (code[:nw].copy(), carr[:nw].copy())
        assert np.abs(code[:nw] - ref_code[:nw]).max() < 2e-6 * np.abs(ref_code).max()
    assert np.array_equal(code[:nw], refs[nw][0]) and np.array_equal(carr[:nw], refs[nw][1]), "banks differ after %d launches" % n
    assert bcs.dev_status() == 0
print("ride stress: %d launches, kernel %s, means / banks identical every time, status 0" % (n, bcs.stage1_kernel))
bcs.Stop()
