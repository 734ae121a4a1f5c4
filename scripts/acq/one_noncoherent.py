import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import navlab_dpe_sdr_amd as dpe
fs, S = 2.5e6, 25000
ch = dpe.synth.random_channels(77, 6, prns=[3, 7, 11, 18, 22, 31]); ch["cp_ref"] = ch["cp"].copy()
iq = dpe.synth.gen_iq(78, fs, S, ch, amp=150.0, flip=np.zeros(6, dtype=bool))
d = torch.from_numpy(iq).to("cuda:0")
bins = (np.arange(125) - 62) * 100.0
acq = dpe.Acquisition(fs, S, list(range(1, 33)), bins, mode="noncoherent", prn_chunk=32)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6): acq.search(d)
torch.cuda.synchronize()
