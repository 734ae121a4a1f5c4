"""Synthetic workloads for bench.py / smoke(): the R-twin of the reference demo.

Channel state comes from the product's cuChanMgr restatement started on the reference's handoff
file (data/handoff_params_usrp6.csv, a copy of demofiles/handoff_params_usrp6.csv) and advanced
open-loop with a static receiver; each window's int16 I/Q is synthesised from that window's
start-referenced channel parameters, so the correlation peaks sit at the grid centre."""
import os

import numpy as np

from . import engine, handoff, synth

HANDOFF_CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "handoff_params_usrp6.csv")

CONFIG_R = dict(name="R: demofiles twin, 2.5 Msps x 20 ms, 8 SVs, rngrid3-format 25^4-point grids",
                fs=2.5e6, S=50000, K=8, G=390625, L=4, B=20, amp=48.0)


CONFIG_H = dict(name="H: synthetic 25 Msps x 20 ms, 12 SVs, 1e5-point rngrid3-format grids",
                fs=25e6, S=500000, K=12, G=100000, L=31, B=16, amp=15.2)

# BASELINE.json configs[3]: 1e6-point grid over the GPUs of one node (strong scaling: the GLOBAL grid is fixed)
CONFIG_M = dict(name="M: synthetic 2.5 Msps x 20 ms, 8 SVs, 1e6-point rngrid3-format grids sharded over the GPUs",
                fs=2.5e6, S=50000, K=8, G=1000000, L=4, B=20, amp=48.0)

_C = 299792458.0
_FCA = 1.023e6
_FL1 = 1.57542e9
_OE = 7.2921151467e-5


def extend_handoff(ho, K_total, elev_min_deg=10.0):
    """Handoff state with K_total SVs: the file's 8 plus synthetic ones (SURVEY.md 8d, K=12 uses PRNs
    1,5,10,25).  A synthetic SV clones a real ephemeris with shifted OMEGA_0 / M_0; its code phase,
    reference code period and Doppler are then derived from the geometry with the product's own
    cuChanMgr restatement so that the extended state is self-consistent at rxTime."""
    K0 = len(ho["prn_list"])
    if K_total <= K0:
        return ho
    out = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in ho.items()}
    X = ho["X_ECEF"]
    R = None
    have = set(int(q) for q in ho["prn_list"])
    extra_prns = [p for p in synth.PRNS_H if p not in have]
    extra_prns += [p for p in range(1, 38) if p not in have and p not in extra_prns]
    add = dict(prn=[], rc=[], ri=[], fc=[], fi=[], cp=[], cpr=[], tow=[], eph=[])
    j = 0
    for trial in range(1, 2000):
        if len(add["prn"]) == K_total - K0:
            break
        eph = ho["eph"][trial % K0].copy()
        eph[3] += 0.9 * trial      # OMEGA_0
        eph[5] += 1.7 * trial      # M_0
        tow = int(ho["TOW"][0])
        cp, rc, cpr, fc, fi = 1000, 0.0, 1000, _FCA, 0.0
        ok = True
        for it in range(4):        # fixed point: transmit time <-> satellite position <-> pseudorange
            cm = engine.ChanMgr([1], [rc], [0.0], [fc], [fi], [cp], [cpr], [tow], eph[None, :], ho["rxTime"], 0.02)
            cm.Start(X, X, (0.0,))
            cm.Update(X, X, (0.0,))
            s, e, w, batch = cm.outputs(with_batch=True)
            cm.Stop()
            sat = e["satState"][0]
            if R is None:
                R = w["enu2ecef"][0].reshape(3, 3)
            los = sat[:3] - X[:3]
            rng = np.linalg.norm(los)
            elev = np.degrees(np.arcsin((R.T @ (los / rng))[2]))
            if elev < elev_min_deg:
                ok = False
                break
            tx = ho["rxTime"] - (rng - _C * sat[3] + X[3]) / _C      # consistent transmit time at rxTime
            whole = np.floor((tx - tow) * 1000.0)
            cpr = int(cp - whole)
            rc = float((tx - tow - whole * 1e-3) * _FCA)
            fc, fi = float(s["codeFrequency"][0]), float(s["carrierFrequency"][0])   # measurement-updated by Update()
        if not ok or not (0.0 <= rc < 1023.0):
            continue
        add["prn"].append(extra_prns[j]); j += 1
        add["rc"].append(rc); add["ri"].append(0.25 * trial % 1.0); add["fc"].append(fc); add["fi"].append(fi)
        add["cp"].append(cp); add["cpr"].append(cpr); add["tow"].append(tow); add["eph"].append(eph)
    if len(add["prn"]) != K_total - K0:
        raise RuntimeError("could not place %d synthetic SVs above the horizon" % (K_total - K0))
    out["prn_list"] = np.concatenate([ho["prn_list"], np.array(add["prn"], dtype=np.int32)])
    for k, a in (("rc", "rc"), ("ri", "ri"), ("fc", "fc"), ("fi", "fi")):
        out[k] = np.concatenate([ho[k], np.array(add[a])])
    out["cp"] = np.concatenate([ho["cp"], np.array(add["cp"], dtype=np.int32)])
    out["cp_timestamp"] = np.concatenate([ho["cp_timestamp"], np.array(add["cpr"], dtype=np.int32)])
    out["TOW"] = np.concatenate([ho["TOW"], np.array(add["tow"], dtype=np.int32)])
    out["eph"] = np.vstack([ho["eph"], np.array(add["eph"])])
    return out


def build_windows(W, fs, S, K, seed=0, amp=48.0, velocity=None, truth_out=None):
    """-> iq int16 [W, 2S], chan_start [W,K], chan_end [W,K], bcm_window [W].
    velocity: ECEF velocity (m/s, 3-vector) of the simulated receiver; None = static at the handoff position.
    truth_out: optional list that receives the true state [8] at the START of every window."""
    ho = extend_handoff(handoff.read_handoff(HANDOFF_CSV), K)
    cm = engine.ChanMgr.from_handoff(ho, S / fs, K)
    X0 = np.array(ho["X_ECEF"], dtype=np.float64)
    iq = np.empty((W, 2 * S), dtype=np.int16)
    cs = np.zeros((W, K), dtype=engine.CHAN_START_DTYPE)
    ce = np.zeros((W, K), dtype=engine.CHAN_END_DTYPE)
    bw = np.zeros(W, dtype=engine.BCM_WINDOW_DTYPE)
    for w in range(W):
        X = X0.copy()
        if velocity is not None:
            X[:3] += np.asarray(velocity, dtype=np.float64) * (S / fs) * w
            X[4:7] = velocity
        if truth_out is not None:
            truth_out.append(X.copy())
        (cm.Start if w == 0 else cm.Update)(X, X, (0.0,))
        s, e, win = cm.outputs()
        cs[w], ce[w], bw[w] = s, e, win[0]
        ch = dict(prn=s["prn"], rc=s["codePhaseStart"], ri=s["carrierPhaseStart"], fc=s["codeFrequency"],
                  fi=s["carrierFrequency"], cp=s["cpElapsedStart"], cp_ref=s["cpReference"])
        iq[w] = synth.gen_iq(seed * 100003 + w, fs, S, ch, amp=amp)
    cm.Stop()
    return iq, cs, ce, bw


def build_grids(G_local, rank=0, world=1, seed=3):
    """This rank's contiguous slice of the global rngrid3-format grids, weak scaling: world x G_local points."""
    pos = synth.rand_grid(seed, G_local * world)
    vel = synth.rand_grid(seed + 1, G_local * world, half=(6.0, 6.0, 6.0, 3.0))
    sl = slice(rank * G_local, (rank + 1) * G_local)
    return pos, vel, pos[sl], vel[sl], rank * G_local


def build_grids_strong(G_global, rank=0, world=1, seed=3):
    """Strong scaling: the global grids are fixed, rank r takes sharding.shard_range(G_global, r, world)."""
    from . import sharding
    pos = synth.rand_grid(seed, G_global)
    vel = synth.rand_grid(seed + 1, G_global, half=(6.0, 6.0, 6.0, 3.0))
    b, e = sharding.shard_range(G_global, rank, world)
    return pos, vel, pos[b:e], vel[b:e], b
