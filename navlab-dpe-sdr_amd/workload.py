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


def build_windows(W, fs, S, K, seed=0, amp=48.0):
    """-> iq int16 [W, 2S], chan_start [W,K], chan_end [W,K], bcm_window [W]."""
    ho = handoff.read_handoff(HANDOFF_CSV)
    cm = engine.ChanMgr.from_handoff(ho, S / fs, K)
    X = ho["X_ECEF"]
    iq = np.empty((W, 2 * S), dtype=np.int16)
    cs = np.zeros((W, K), dtype=engine.CHAN_START_DTYPE)
    ce = np.zeros((W, K), dtype=engine.CHAN_END_DTYPE)
    bw = np.zeros(W, dtype=engine.BCM_WINDOW_DTYPE)
    for w in range(W):
        (cm.Start if w == 0 else cm.Update)(X, X, (0.0,))
        s, e, win = cm.outputs()
        cs[w], ce[w], bw[w] = s, e, win[0]
        ch = dict(prn=s["prn"], rc=s["codePhaseStart"], ri=s["carrierPhaseStart"], fc=s["codeFrequency"],
                  fi=s["carrierFrequency"], cp=s["cpElapsedStart"], cp_ref=s["cpReference"])
        iq[w] = synth.gen_iq(seed * 100003 + w, fs, S, ch, amp=amp)
    cm.Stop()
    return iq, cs, ce, bw


def build_grids(G_local, rank=0, world=1, seed=3):
    """This rank's contiguous slice of the global rngrid3-format grids."""
    pos = synth.rand_grid(seed, G_local * world)
    vel = synth.rand_grid(seed + 1, G_local * world, half=(6.0, 6.0, 6.0, 3.0))
    sl = slice(rank * G_local, (rank + 1) * G_local)
    return pos, vel, pos[sl], vel[sl], rank * G_local
