"""Shared test/bench-checker helpers: build seeded cases, run the oracle (CPU) and the HIP path
(through the C-ABI) on the same inputs, compare.  Uses oracle/ -- test infrastructure only."""
import os

import numpy as np

import navlab_dpe_sdr_amd as dpe

HERE = os.path.dirname(os.path.abspath(__file__))
HANDOFF = dpe.workload.HANDOFF_CSV


def _oracle():
    from oracle import oracle as o
    return o


def make_case(seed=0, fs=2.5e6, S=50000, K=8, G=4096, amp=200.0, W=1, grid="rand", vel_G=None,
              center_offset=None, flips=None):
    """W consecutive windows on the handoff geometry (static receiver), channel state advanced by
    the oracle's cuChanMgr restatement; I/Q synthesised from each window's start-referenced params."""
    o = _oracle()
    ho = dpe.workload.extend_handoff(dpe.handoff.read_handoff(HANDOFF), K)   # K > 8: synthetic extra SVs
    T = S / fs
    X = ho["X_ECEF"].copy()
    sl = slice(0, K)
    cm = o.ChanMgr(ho["prn_list"][sl], ho["rc"][sl], ho["ri"][sl], ho["fc"][sl], ho["fi"][sl], ho["cp"][sl],
                   ho["cp_timestamp"][sl], ho["TOW"][sl], ho["eph"][sl], ho["rxTime"], T)
    if grid == "rand":
        pos = dpe.synth.rand_grid(seed + 100, G)
        vel = dpe.synth.rand_grid(seed + 200, vel_G or G, half=(6.0, 6.0, 6.0, 3.0))
    elif grid == "uniform":
        dim = int(round(G ** 0.25))
        pos = dpe.synth.uniform_grid(dim, 1.0)
        vel = dpe.synth.uniform_grid(dim, 1.0)
    elif grid == "spread":
        pos, vel = dpe.synth.spread_grid()
    else:
        raise ValueError(grid)
    tg = np.zeros(1)
    centre = X.copy()
    if center_offset is not None:
        R0 = o.enu2ecef(o.ecef2ll(X)).reshape(3, 3)
        centre[:3] += R0 @ np.asarray(center_offset[:3], dtype=np.float64)
        centre[3] += center_offset[3]
    wins = []
    rng = np.random.Generator(np.random.PCG64(seed))
    for w in range(W):
        batch, R = (cm.start(X, centre, tg) if w == 0 else cm.update(X, centre, tg))
        start = dict(prn=cm.prns, rc=cm.rcStart.copy(), ri=cm.riStart.copy(), fc=cm.fc.copy(), fi=cm.fi.copy(),
                     cp=cm.cpElaStart.copy(), cp_ref=cm.cpRef.copy())
        fl = rng.integers(0, 2, K).astype(bool) if flips is None else np.asarray(flips, dtype=bool)
        iq = dpe.synth.gen_iq(seed * 1000 + w, fs, S, start, amp=amp, flip=fl)
        wins.append(dict(iq=iq, start=start, sat=batch[:, 0].copy(), R=R.copy(), rxTime=cm.rxTime,
                         rcEnd=cm.rcEnd.copy(), cpElaEnd=cm.cpElaEnd.copy(), cpRef=cm.cpRef.copy(),
                         cpRefTOW=cm.cpRefTOW.copy(), fc=cm.fc.copy(), fi=cm.fi.copy(), centre=centre.copy(),
                         flip=fl))
    return dict(fs=fs, S=S, K=K, W=W, C=dpe.engine.carr_fft_len(S), pos=pos, vel=vel, wins=wins,
                prn=np.asarray(cm.prns))


def run_oracle(case, L, B, lpower=1, windows=None):
    o = _oracle()
    fs, S, K, C = case["fs"], case["S"], case["K"], case["C"]
    out = dict(code=[], carr=[], pos=[], pos_x=[], vel=[], res=[], info=[])
    for wi, w in enumerate(case["wins"]):
        if windows is not None and wi not in windows:
            continue
        code, carr, info = [], [], []
        for k in range(K):
            s = w["start"]
            c, f, inf = o.bcs_sv(w["iq"], fs, int(s["prn"][k]), s["rc"][k], s["ri"][k], s["fc"][k], s["fi"][k],
                                 int(s["cp"][k]), int(s["cp_ref"][k]), -L, L, -B, B, C)
            code.append(c)
            carr.append(f)
            info.append(inf)
        code, carr = np.stack(code), np.stack(carr)
        sp, oobp = o.bcm_pos(w["sat"], code, S // 2 - L, w["centre"], case["pos"], w["R"], w["fc"], w["cpRefTOW"],
                             w["cpElaEnd"], w["cpRef"], w["rcEnd"], w["rxTime"], fs, S, lpower)
        quirks = o.bcm_pos_quirks()
        spx, oobx = o.bcm_pos(w["sat"], code, S // 2 - L, w["centre"], case["pos"], w["R"], w["fc"], w["cpRefTOW"],
                           w["cpElaEnd"], w["cpRef"], w["rcEnd"], w["rxTime"], fs, S, lpower, extended=True)
        sv, oobv = o.bcm_vel(w["sat"], carr, C // 2 - B, w["centre"], case["vel"], w["R"], w["fi"], w["rxTime"], fs,
                             C, 1, lpower)
        ip, iv = o.argmax_first(sp), o.argmax_first(sv)
        z, _ = o.make_meas(ip, iv, w["centre"], case["pos"], case["vel"], w["R"])
        out["code"].append(code)
        out["carr"].append(carr)
        out["pos"].append(sp)
        out["pos_x"].append(spx)
        out["vel"].append(sv)
        out["info"].append(info)
        out["res"].append(dict(posIndex=ip, velIndex=iv, zVal=z, posOutOfWindow=oobp, velOutOfWindow=oobv,
                               posOutOfWindowX=oobx))
        out.setdefault("pos_quirk", []).append(quirks)
        out.setdefault("R", []).append(w["R"])
        out.setdefault("centre", []).append(w["centre"])
    out["pos_grid"], out["vel_grid"] = case["pos"], case["vel"]
    return out


def pack_gpu_inputs(case):
    """Numpy side of the C-ABI inputs for all windows of a case."""
    wins = case["wins"]
    cs = np.stack([dpe.engine.chan_start_array(w["start"]["prn"], w["start"]["rc"], w["start"]["ri"], w["start"]["fc"],
                                               w["start"]["fi"], w["start"]["cp"], w["start"]["cp_ref"]) for w in wins])
    ce = np.stack([dpe.engine.chan_end_array(w["sat"], w["rcEnd"], w["fc"], w["fi"], w["cpRefTOW"], w["cpElaEnd"],
                                             w["cpRef"]) for w in wins])
    bw = np.concatenate([dpe.engine.bcm_window_array(w["centre"][None, :], w["R"][None, :], [w["rxTime"]]) for w in wins])
    iq = np.stack([w["iq"] for w in wins])
    return iq, cs, ce, bw


def run_gpu(case, L, B, lpower=1, write_scores=True, weighted_mean=True, repeats=1, reference_pair=False):
    import torch
    iq, cs, ce, bw = pack_gpu_inputs(case)
    W, K = cs.shape
    dev = torch.device("cuda:0")
    iq_d = torch.from_numpy(iq).to(dev)
    bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=case["S"], lag_half_width=L, bin_half_width=B,
                              max_windows=W, max_channels=K)
    bcs.Start()
    bcm = dpe.BatchCorrManifold(case["fs"], case["S"], bcs.NumFFTPoints, case["pos"], case["vel"], LPower=lpower,
                                lag_half_width=L, bin_half_width=B, max_windows=W, max_channels=K,
                                write_scores=write_scores, weighted_mean=weighted_mean, reference_pair=reference_pair)
    bcm.Start()
    for _ in range(repeats):          # same handles, back-to-back Updates (staging ring, alternating key sets)
        bcs.Update(iq_d, cs)
        bcm.Update(bcs.CodeScores, bcs.CarrScores, bw, ce)
    res = bcm.results()
    code, carr = bcs.read_banks()
    idx_next, no_flip, mean = bcs.read_info()
    out = dict(code=list(code), carr=list(carr), res=res, idx_next=idx_next, no_flip=no_flip, mean=mean)
    if write_scores:
        ps, vs = bcm.read_scores()
        out["pos"], out["vel"] = list(ps), list(vs)
    bcm.Stop()
    bcs.Stop()
    return out


POS_REF_NOISE = 1e-4


def assert_parity(gpu, ref, tol=2e-5, check_scores=True, pos_ref_noise=None):
    """fp32 HIP path vs fp64 oracle.  Tolerances are relative to the peak magnitude / max score.

    Position-manifold scores are held to `tol` against the oracle's extended-precision index
    (pos_x) and to POS_REF_NOISE against the faithful fp64 restatement: the reference's own
    `rxTime - pr/C` (batchcorrmanifold.cu:1784, rxTime ~4e5 s) rounds at 5.8e-11 s = 17 mm =
    1.4e-4 samples per (point, SV), i.e. ~3e-5 of the score -- measured, see DESIGN.md."""
    n = len(ref["res"])
    pos_noise = POS_REF_NOISE if pos_ref_noise is None else pos_ref_noise
    worst = dict(code=0.0, carr=0.0, pos=0.0, vel=0.0)
    for w in range(n):
        for name in ("code", "carr"):
            r, g = ref[name][w], gpu[name][w]
            err = np.abs(g - r).max() / np.abs(r).max()
            worst[name] = max(worst[name], err)
            assert err < tol, "%s bank window %d: rel err %.3g" % (name, w, err)
        for k, inf in enumerate(ref["info"][w]):
            assert gpu["idx_next"][w, k] == inf["idx_next"]
            assert bool(gpu["no_flip"][w, k]) == inf["no_flip_larger"]
        rm = ref["info"][w][0]["mean"]
        assert gpu["mean"][w] == rm, "DC mean must be bit-exact (integer sums)"
        if check_scores:
            for name, rname, lim in (("pos", "pos_x", tol), ("pos", "pos", pos_noise), ("vel", "vel", tol)):
                r, g = ref[rname][w], gpu[name][w]
                if rname == "pos":
                    keep = np.ones(r.size, dtype=bool)
                    # points where the reference's floor(idx) / floor(idx+1) pair double-counts (oracle/dpe_oracle.c):
                    # its value there is decided by the last bit of its fp64 index; held to the continuous value only
                    keep[ref["pos_quirk"][w]] = False
                    if ref["res"][w]["posOutOfWindow"] > 0:
                        # banks narrower than the grid reaches: a pair within the reference's own index noise of the
                        # bank edge is dropped by one evaluation and kept by the other; a bounded number of such points
                        # is set aside (they are still held to `tol` against the extended-precision index above)
                        flips = np.abs(r - ref["pos_x"][w]) > 10 * pos_noise * r.max()
                        # expected number: (pairs per sample of index) x 2 edges x the reference's 1.4e-4-sample noise
                        allowed = 16 + r.size * len(ref["info"][w]) // 2000
                        assert (flips & keep).sum() <= allowed, "too many edge flips between the faithful and the extended index"
                        keep &= ~flips
                    r, g = r[keep], g[keep]
                    if r.size == 0:
                        continue
                if r.max() == 0:          # every pair of every point outside the banks: all scores exactly 0
                    assert not g.any()
                    continue
                oob_here = ref["res"][w]["velOutOfWindow" if name == "vel" else "posOutOfWindowX"]
                if rname != "pos" and oob_here > 0:
                    # truncated banks are this build's device, not the reference's: a pair whose index sits within fp32
                    # rounding of a bank edge may be kept here and dropped by the fp64 oracle (or vice versa) -- a whole
                    # SV contribution at that point.  At most two such points per window are set aside (seen: 1 in ~3000
                    # narrow-bank cases); the out-of-window counts below then differ by as many.
                    d = np.abs(g - r) / r.max()
                    edge = np.argsort(-d)[:2]
                    edge = edge[d[edge] > 100 * lim]
                    if edge.size:
                        m = np.ones(r.size, dtype=bool)
                        m[edge] = False
                        r, g = r[m], g[m]
                        if r.size == 0 or r.max() == 0:
                            continue
                err = np.abs(g - r).max() / r.max()
                worst[rname] = max(worst.get(rname, 0.0), err)
                assert err < lim, "%s scores vs %s window %d: rel err %.3g" % (name, rname, w, err)
        rr, gr = ref["res"][w], gpu["res"][w]
        for name, key in (("pos", "posIndex"), ("vel", "velIndex")):
            if gr[key] != rr[key]:   # only acceptable as an fp32 tie
                r, best = ref[name][w], rr[key]
                if name == "pos" and len(ref["pos_quirk"][w]):
                    r = ref["pos_x"][w]          # the faithful arg-max may sit on a double-counted point
                    best = int(np.argmax(r))
                lim = pos_noise if name == "pos" else tol
                assert abs(r[gr[key]] - r[best]) <= lim * r.max(), "%s arg-max differs beyond tolerance" % name
        if gr["posIndex"] == rr["posIndex"] and gr["velIndex"] == rr["velIndex"]:
            assert np.abs(gr["zVal"] - rr["zVal"]).max() < 1e-6     # same grid point -> same fix
        # pairs outside the banks: exact against the extended-precision index; the faithful fp64 index moves a few
        # pairs at the window edges by its own rxTime - pr/C rounding (1.4e-4 samples) when the banks are narrow
        assert abs(gr["posOutOfWindow"] - rr["posOutOfWindowX"]) <= (2 if rr["posOutOfWindowX"] else 0)
        assert abs(gr["velOutOfWindow"] - rr["velOutOfWindow"]) <= (2 if rr["velOutOfWindow"] else 0)
        assert abs(gr["posOutOfWindow"] - rr["posOutOfWindow"]) <= 8 + ref["pos"][w].size * len(ref["info"][w]) // 2000
        if ref["pos_x"][w].sum() == 0 or ref["vel"][w].sum() == 0:
            continue    # a manifold whose every score is 0 has no weighted mean (0/0, NaN here as in the reference)
        if "zValMean" in gr and "pos_grid" in ref and np.any(gr["zValMean"] != 0):   # "Method 1" weighted-mean estimator vs fp64 sums of the oracle scores
            zp = (ref["pos_x"][w][:, None] * ref["pos_grid"]).sum(0) / ref["pos_x"][w].sum()
            zv = (ref["vel"][w][:, None] * ref["vel_grid"]).sum(0) / ref["vel"][w].sum()
            R = ref["R"][w].reshape(3, 3)
            c = ref["centre"][w]
            zm = np.concatenate([R @ zp[:3] + c[:3], [zp[3] + c[3]], R @ zv[:3] + c[4:7], [zv[3] + c[7]]])
            # Truncated banks (pairs outside the window): up to two points per window may keep / drop a whole SV contribution at a bank
            # edge against the fp64 oracle (set aside in the score check above); each moves the weighted mean by at most its share of the
            # score sum times the grid's extent.  (Found by the sweep at DPE_FUZZ_SEED=777, case 250: L = 1, 2 204 points.)
            ep = 2 * ref["pos_x"][w].max() / ref["pos_x"][w].sum() * 2 * np.abs(ref["pos_grid"]).max() if rr["posOutOfWindowX"] > 0 else 0.0
            ev = 2 * ref["vel"][w].max() / ref["vel"][w].sum() * 2 * np.abs(ref["vel_grid"]).max() if rr["velOutOfWindow"] > 0 else 0.0
            assert np.abs(gr["zValMean"][:4] - zm[:4]).max() < 1e-3 + ep and np.abs(gr["zValMean"][4:] - zm[4:]).max() < 1e-4 + ev
    return worst
