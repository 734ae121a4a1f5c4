"""A/B: non-coherent 32 x 125 search time under env variants (each in a child process)."""
import os, sys, subprocess, json
CHILD = r'''
import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
fs, S = 2.5e6, 25000
ch = dpe.synth.random_channels(77, 6, prns=[3, 7, 11, 18, 22, 31]); ch["cp_ref"] = ch["cp"].copy()
iq = dpe.synth.gen_iq(78, fs, S, ch, amp=150.0, flip=np.zeros(6, dtype=bool))
d = torch.from_numpy(iq).to("cuda:0")
for nb, step in ((125, 100.0), (25, 500.0)):
    bins = (np.arange(nb) - nb // 2) * step
    acq = dpe.Acquisition(fs, S, list(range(1, 33)), bins, mode="noncoherent", prn_chunk=32)
    for _ in range(10): acq.search(d)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t = dpe.engine.HipEventTimer(); t.start()
        for _ in range(40): acq.search(d)
        t.stop(); best = min(best, t.elapsed_ms() / 40)
    res = acq.results()
    print("bins", nb, "ms %.4f" % best, "found", sorted(r["prn"] for r in res if r["found"]))
'''
variants = [("pack+fwd", {})] + [(a, dict(kv.split("=") for kv in a.split(","))) for a in sys.argv[1:]]
for rep in range(2):
    for name, env in variants:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True, timeout=600)
        print(name, "|", " | ".join(r.stdout.strip().split("\n")), r.stderr[-300:] if r.returncode else "")
