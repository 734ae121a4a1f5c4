#!/usr/bin/env python3
"""Coarse-acquisition timing on one MI355X (BASELINE.json configs[4] shape): 32 PRNs x 125 Doppler bins
x all 2500 code delays of a 10 ms / 2.5 Msps window.  Prints one JSON line: search cells per second
(PRN x bin x delay), ms per window, and the CPU oracle (numpy fp64 restatement of the reference's
coarse_acquisition) on a bounded sample next to it."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import navlab_dpe_sdr_amd as dpe
    mode = sys.argv[1] if len(sys.argv) > 1 else "coherent"
    fs, S = 2.5e6, 25000
    ch = dpe.synth.random_channels(77, 6, prns=[3, 7, 11, 18, 22, 31])
    ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(78, fs, S, ch, amp=150.0, flip=np.zeros(6, dtype=bool))
    bins = np.arange(-62, 63) * 100.0
    prns = list(range(1, 33))
    acq = dpe.Acquisition(fs, S, prns, bins, mode=mode, prn_chunk=32)
    d = torch.from_numpy(iq).to("cuda:0")
    for _ in range(3):
        acq.search(d)
    torch.cuda.synchronize()
    t = dpe.engine.HipEventTimer()
    n = 20
    t.start()
    for _ in range(n):
        acq.search(d)
    t.stop()
    ms = t.elapsed_ms() / n
    res = acq.results()
    acq.search_signal(d)
    t0 = time.perf_counter()
    for _ in range(5):
        full = acq.search_signal(d)                    # coarse + fine frequency, host-synchronous
    ms_full = (time.perf_counter() - t0) / 5 * 1e3
    t0 = time.perf_counter()
    for _ in range(5):
        acq.search(d); coarse = acq.results()
    ms_coarse = (time.perf_counter() - t0) / 5 * 1e3
    cells = len(prns) * bins.size * (S // 10)
    out = {"metric": "acquisition search cells (PRN x Doppler bin x code delay) per second", "mode": mode,
           "value": cells / (ms * 1e-3), "ms_per_window": ms, "x_realtime": 10.0 / ms,
           "search_signal_ms_per_window": ms_full, "search_plus_results_ms": ms_coarse,
           "found": sorted(r["prn"] for r in res if r["found"]), "truth": sorted(int(p) for p in ch["prn"])}
    from oracle import oracle as o
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 8.0:
        o.coarse_acquisition(iq, fs, prns[k % 32], bins, coherent=(mode == "coherent"), mode="textbook" if mode == "textbook" else None)
        k += 1
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": k * bins.size * (S // 10) / dt, "cores": 1, "kind": "port", "sample": "%d PRNs, %.1f s" % (k, dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
