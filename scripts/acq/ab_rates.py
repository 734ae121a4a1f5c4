"""A/B: coherent / textbook 32 x 125 search at 4 and 5 Msps, fused vs rocFFT chain."""
import os, sys, subprocess
CHILD = r'''
import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
for fs in (4.0e6, 5.0e6):
    S = int(fs * 1e-2)
    ch = dpe.synth.random_channels(77, 6, prns=[3, 7, 11, 18, 22, 31]); ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(78, fs, S, ch, amp=150.0, flip=np.zeros(6, dtype=bool))
    d = torch.from_numpy(iq).to("cuda:0")
    bins = (np.arange(125) - 62) * 100.0
    for mode in ("coherent", "textbook"):
        acq = dpe.Acquisition(fs, S, list(range(1, 33)), bins, mode=mode, prn_chunk=32 if mode == "coherent" else 8)
        for _ in range(10): acq.search(d)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t = dpe.engine.HipEventTimer(); t.start()
            for _ in range(10): acq.search(d)
            t.stop(); best = min(best, t.elapsed_ms() / 10)
        res = acq.results()
        print("%.0f Msps %s ms %.4f found %s" % (fs / 1e6, mode, best, sorted(r["prn"] for r in res if r["found"])))
        acq.close()
'''
variants = [("bpb_default", {}), ("bpb8", {"DPE_ACQ_BPB": "8"}), ("bpb4", {"DPE_ACQ_BPB": "4"}), ("bpb2", {"DPE_ACQ_BPB": "2"}), ("bpb5", {"DPE_ACQ_BPB": "5"})]
for name, env in variants:
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=900)
    print(name, "|", " | ".join(r.stdout.strip().split("\n")), r.stderr[-400:] if r.returncode else "")
