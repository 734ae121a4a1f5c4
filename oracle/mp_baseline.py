"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Timing harness of the oracle for bench.py's `cpu_baseline` leg.

`full_window` is one whole 20 ms window through the fp64 restatement: FFT BatchCorrScores per SV
(numpy, cudarecv/modules/src/batchcorrscores.cu:1043-1180) + both manifold scans and arg-max in C
(batchcorrmanifold.cu:1710-1828, 1861-1963, 2589-2590).  Run as a program it is one worker of the
all-cores measurement ("single-thread and over all host cores"): windows are independent, so
N processes each loop over whole windows of the same workload between a file barrier and a deadline.

    python -m oracle.mp_baseline <dir> <worker index> <seconds>
"""
import json
import os
import sys
import time

import numpy as np

from . import oracle as o


def carr_fft_len(S):
    return o.carr_fft_len(S)


def full_window(d, w):
    """d: dict of the workload arrays (see save_workload); w: window index."""
    fs, S, L, B = float(d["fs"]), int(d["S"]), int(d["L"]), int(d["B"])
    iq, cs, ce, bw, pos, vel = d["iq"], d["cs"], d["ce"], d["bw"], d["pos"], d["vel"]
    K = cs.shape[1]
    C = o.carr_fft_len(S)
    code, carr = [], []
    for k in range(K):
        c = cs[w, k]
        cc, cf, _ = o.bcs_sv_fft(iq[w], fs, int(c["prn"]), c["codePhaseStart"], c["carrierPhaseStart"],
                                 c["codeFrequency"], c["carrierFrequency"], int(c["cpElapsedStart"]),
                                 int(c["cpReference"]))
        code.append(cc[S // 2 - L:S // 2 + L + 1])
        carr.append(cf[C // 2 - B:C // 2 + B + 1])
    e = ce[w]
    sp, _ = o.bcm_pos(e["satState"], np.stack(code), S // 2 - L, bw[w]["xCurrkk1"], pos, bw[w]["enu2ecef"],
                      e["codeFrequency"], e["cpRefTOW"], e["cpElapsedEnd"], e["cpRef"], e["codePhaseEnd"],
                      float(bw[w]["rxTime"]), fs, S, 1)
    sv, _ = o.bcm_vel(e["satState"], np.stack(carr), C // 2 - B, bw[w]["xCurrkk1"], vel, bw[w]["enu2ecef"],
                      e["carrierFrequency"], float(bw[w]["rxTime"]), fs, C, 1, 1)
    return o.argmax_first(sp), o.argmax_first(sv)


def save_workload(path, **arrays):
    np.savez(path, **arrays)


def load_workload(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    flags = "gcc -O2 -std=c11 -ffp-contract=off (oracle/Makefile); numpy %s pocketfft" % np.__version__
    return {"nproc": os.cpu_count(), "usable": len(os.sched_getaffinity(0)), "cpu": model, "build": flags}


def worker(rundir, index, seconds):
    d = load_workload(os.path.join(rundir, "workload.npz"))
    o.lib()
    nw = d["iq"].shape[0]
    full_window(d, 0)                                    # page everything in
    open(os.path.join(rundir, "ready.%d" % index), "w").close()
    go = os.path.join(rundir, "go")
    t_wait = time.time()
    while not os.path.exists(go):
        if time.time() - t_wait > 300:
            return 1
        time.sleep(0.005)
    n, t0 = 0, time.perf_counter()
    while True:
        full_window(d, n % nw)
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds:
            break
    print(json.dumps({"n": n, "dt": dt}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(worker(sys.argv[1], int(sys.argv[2]), float(sys.argv[3])))
