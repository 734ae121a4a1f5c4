import sys, numpy as np, torch, ctypes as C
sys.path.insert(0, '.')
import navlab_dpe_sdr_amd as dpe
from tests import helpers
from oracle import oracle as o
case = helpers.make_case(seed=2, S=50000, K=8, G=20000, amp=200.0, grid="spread")
L, B = 8, 48
ref = helpers.run_oracle(case, L, B)
out = helpers.run_gpu(case, L, B)
for name in ("code", "carr"):
    r, g = ref[name][0], out[name][0]
    print(name, "per-SV rel err", np.abs(g - r).max(axis=1) / np.abs(r).max(axis=1))
for name in ("pos", "vel"):
    r, g = ref[name][0], out[name][0]
    print(name, "score rel err", np.abs(g - r).max() / r.max(), "argmax", g.argmax(), r.argmax())
# BCM alone with oracle banks
iq, cs, ce, bw = helpers.pack_gpu_inputs(case)
K = 8
bcm = dpe.BatchCorrManifold(case["fs"], case["S"], case["C"], case["pos"], case["vel"], lag_half_width=L, bin_half_width=B, max_channels=K)
bcm.Start()
code = torch.from_numpy(ref["code"][0].astype(np.complex64)).to("cuda:0")
carr = torch.from_numpy(ref["carr"][0].astype(np.complex64)).to("cuda:0")
bcm.Update(code, carr, bw, ce)
ps, vs = bcm.read_scores()
print("BCM-only pos rel err", np.abs(ps[0] - ref["pos"][0]).max() / ref["pos"][0].max())
print("BCM-only vel rel err", np.abs(vs[0] - ref["vel"][0]).max() / ref["vel"][0].max())
e = np.abs(ps[0] - ref["pos"][0]); i = e.argmax(); print("worst pos point", i, case["pos"][i], ps[0][i], ref["pos"][0][i])
