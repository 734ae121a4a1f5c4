#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE's own Python twin.

Run in the dev container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

What it does: copies /root/reference/pygnss/pythonreceiver to a scratch dir under /tmp,
converts it with lib2to3 (the reference is Python 2.7), sets numpy.mat = numpy.asmatrix
(removed in NumPy 2), imports it from there and records input/output vectors of the
reference callables that the CUDA authors used as their oracle:

  O1 Correlator._make_L1_CAcode_chips            correlator.py:474-515
  O3 Correlator.vector_correlate_unfolded        correlator.py:367-465
  O4 satpos.satellite_clock_correction/locate_satellite   satpos.py:8-185
  O5 utils.ECEF_to_LLA / ECEF_to_ENU rotation    utils.py:13-81,235-275
  O6 NavigationGuesses.generate_spread_grid      receiver.py:995-1026
  O7 Receiver.dp_track internals (one iteration) receiver.py:205-397, channel.py:194-245
  O8 Correlator.coarse_acquisition               correlator.py:53-103
  O11 libgnss/rinex.py parse_rinex on an excerpt of the reference's demofiles/nist1860.18n (data file)
  O10 ExtendedKalmanFilter._time_update_m5 / _measurement_update_m5 (the real filter, vector/ekf.py:160-178)
  O9 Correlator.search_signal (coarse + fine_frequency_acquisition) on two consecutive windows and
     Receiver.scalar_acquisition's keep-the-better rule   correlator.py:38-51,105-133; receiver.py:452-520

Only DATA (inputs + expected outputs) is written; no reference source is copied into the
repo.  The two harness adaptations below are marked HARNESS and do not touch arithmetic.
"""
import os
import shutil
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SCRATCH = "/tmp/dpe_golden_pygnss"

sys.path.insert(0, ROOT)
import navlab_dpe_sdr_amd as dpe  # noqa: E402  (product-side synthetic generator + handoff reader)


def import_pygnss():
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(SCRATCH)
    shutil.copytree(os.path.join(REF, "pygnss", "pythonreceiver"), os.path.join(SCRATCH, "pythonreceiver"))
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", "pythonreceiver"], cwd=SCRATCH,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    np.mat = np.asmatrix  # HARNESS: removed in NumPy 2
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, SCRATCH)
    from pythonreceiver import receiver
    from pythonreceiver.libgnss import rawfile, satpos, utils
    from pythonreceiver.scalar import correlator
    from pythonreceiver.vector import ekf
    return types.SimpleNamespace(receiver=receiver, rawfile=rawfile, satpos=satpos, utils=utils,
                                 correlator=correlator, ekf=ekf)


class Eph:
    """Container with the attribute names pygnss' libgnss.ephemeris.Ephemerides exposes."""

    def __init__(self, ho, k):
        for j, name in enumerate(dpe.handoff.EPH_FIELDS):
            setattr(self, name, float(ho["eph"][k, j]))
        self.timestamp = {"cp": float(ho["cp_timestamp"][k]), "TOW": float(ho["TOW"][k])}


def open_rawfile(pg, path, fs, T):
    dt = np.dtype([("i", np.short), ("q", np.short)])
    rf = pg.rawfile.RawFile(abspath=path, fs=fs, fi=0.0, ds=1.0, datatype=dt, notes="synthetic", verbose=False)
    # HARNESS: under py3 dict_keys != list, so the reference's format dispatch falls through
    rf.format_rawsnippet = rf.format_rawsnippet_datatype_complex
    rf.set_rawsnippet_settings(T=T, T_big=T, verbose=False)
    return rf


def make_o9(pg):
    """O9: the reference's own two-window acquisition driver on 20 ms of seeded synthetic samples."""
    fs, T = 2.5e6, 0.01
    S = int(round(fs * T))
    ch = dpe.synth.random_channels(9, 4, prns=[3, 11, 22, 31])
    ch["fi"] = np.array([-3671.0, 842.0, 2955.0, -120.0])
    ch["fc"] = 1.023e6 * (1.0 + ch["fi"] / 1.57542e9)
    ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(47, fs, 2 * S, ch, amp=np.array([110.0, 70.0, 140.0, 95.0]), flip=np.zeros(4, dtype=bool))
    path = os.path.join(SCRATCH, "o9.dat")
    iq.tofile(path)
    rf = open_rawfile(pg, path, fs, T)

    class Py2Int(int):           # HARNESS: Python-2 "int / int" floors (correlator.py:78 reshapes by S/N)
        def __truediv__(self, other):
            return int(self) // other
    orig = rf.set_rawsnippet_settings

    def settings(T, T_big, verbose=True):   # HARNESS: keep S an int that floors under "/" after every re-setting
        orig(T=T, T_big=T_big, verbose=False)
        rf.S = Py2Int(rf.S)
    rf.set_rawsnippet_settings = settings
    prn_list = [3, 11, 22, 31, 7, 26]                               # four present, two absent
    # per-window outputs of search_signal (coarse + fine), the two windows scalar_acquisition reads
    rf.set_rawsnippet_settings(T=T, T_big=T)
    per_window = []
    for w in range(2):
        rf.update_rawsnippet()
        rows = []
        for prn in prn_list:
            cor = pg.correlator.Correlator(prn)
            _, found, rc, ri, fc, fi, cppr, cppm = cor.search_signal(rf)
            rows.append([float(found), rc, ri, fc, fi, cppr, cppm])
        per_window.append(rows)
    rf.seek_rawfile(-2 * int(rf.S))
    # the driver itself
    rx = pg.receiver.Receiver(rf, mcount_max=16)
    rx.add_channels(prn_list)
    rx.scalar_acquisition(prn_list, T=T)
    mc = rx._mcount
    final = np.array([[rx.channels[p].rc[mc], rx.channels[p].ri[mc], rx.channels[p].fc[mc], rx.channels[p].fi[mc]]
                      for p in prn_list])
    np.savez_compressed(os.path.join(HERE, "o9_scalar_acquisition.npz"), iq=iq, fs=fs, T=T, S=S,
                        prn_list=np.array(prn_list), per_window=np.array(per_window), final=final,
                        carr_fftpts=int(rf.carr_fftpts),
                        bins=np.asarray(pg.correlator.DOPPLER_SEARCH_MATRIX_COHERENT).ravel(),
                        truth_prn=ch["prn"], truth_rc=ch["rc"], truth_fi=ch["fi"], truth_ri=ch["ri"])
    rf.close_rawfile()


def make_o10(pg):
    """O10: the reference's own Kalman-filter steps (`_m5` methods, i.e. the filter that its `_l5` pass-through
    stands in for) over 40 windows of a synthetic moving fix; F = I as shipped and F = I + T as in CUDARecv."""
    rng = np.random.default_rng(1010)
    x0 = np.array([-2.7e6, -4.29e6, 3.85e6, 120.0, 3.0, -4.0, 1.5, 0.2])
    T = 0.02
    out = {}
    for tag, couple in (("fI", False), ("fT", True)):
        kf = pg.ekf.ExtendedKalmanFilter(np.matrix(x0).T, T=T)
        if couple:   # CUDARecv's transition matrix (cuekf.cu:111-143); a plain parameter of the PyGNSS class
            F = np.eye(8)
            for j in range(4):
                F[j, j + 4] = T
            kf.F = np.asmatrix(F)
        xs, Ps, Qs, Ks, zs, xp, Pp = [], [], [], [], [], [], []
        truth = x0.copy()
        rng2 = np.random.default_rng(7)
        for it in range(40):
            kf._time_update_m5()
            xp.append(np.asarray(kf.X_ECEF).ravel().copy()); Pp.append(np.asarray(kf.Sigma).copy()); Qs.append(np.asarray(kf.Q).copy())
            truth[:4] += truth[4:] * T
            truth[4:7] += rng2.normal(0, 0.3, 3) * (1 + it / 10.0)           # speed sweeps through the Q clamp range
            z = truth + rng2.normal(0, [1, 1, 1, 2, 0.5, 0.5, 0.5, 0.1])
            zs.append(z.copy())
            e = np.matrix(z).T - kf.H * kf.X_ECEF
            kf._measurement_update_m5(e)
            xs.append(np.asarray(kf.X_ECEF).ravel().copy()); Ps.append(np.asarray(kf.Sigma).copy()); Ks.append(np.asarray(kf.K).copy())
        for k, v in (("x_upd", xs), ("P_upd", Ps), ("Q", Qs), ("K", Ks), ("z", zs), ("x_pred", xp), ("P_pred", Pp)):
            out[tag + "_" + k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, "o10_ekf.npz"), x0=x0, T=T, **out)


def make_o11(pg):
    """O11: an excerpt of the reference's RINEX navigation DATA file (header, the stale 2015 record it starts with,
    and every record of the handoff PRNs) + what the reference's Python parser returns for each of those PRNs
    (its rule: the FIRST record of the PRN in the file)."""
    from pythonreceiver.libgnss import rinex as pr
    src = os.path.join(REF, "demofiles", "nist1860.18n")
    ho = dpe.handoff.read_handoff(os.path.join(REF, "demofiles", "handoff_params_usrp6.csv"))
    keep = set(int(p) for p in ho["prn_list"]) | {4}
    with open(src) as f:
        lines = f.read().splitlines()
    h = next(i for i, l in enumerate(lines) if "END OF HEADER" in l) + 1
    out = lines[:h]
    for i in range(h, len(lines) - 7, 8):
        if int(lines[i][0:2]) in keep:
            out += lines[i:i + 8]
    dst = os.path.join(HERE, "o11_nist1860_excerpt.18n")
    with open(dst, "w") as f:
        f.write("\n".join(out) + "\n")
    prns = sorted(keep)
    exp = np.zeros((len(prns), len(dpe.handoff.EPH_FIELDS)))
    for k, p in enumerate(prns):
        e = pr.parse_rinex(dst, p)
        exp[k] = [float(getattr(e, name)) for name in dpe.handoff.EPH_FIELDS]
    np.savez_compressed(os.path.join(HERE, "o11_rinex.npz"), prns=np.array(prns), eph_first=exp,
                        handoff_prns=ho["prn_list"], handoff_eph=ho["eph"], handoff_rxTime=ho["rxTime"])


# ---- O3 / O12: vector_correlate_unfolded on seeded synthetic windows
def run_o3(pg, tag, fs, T, seed, ch, amp, flip, prefix="o3"):
    S = int(round(T * fs))
    iq = dpe.synth.gen_iq(seed, fs, S, ch, amp=amp, flip=flip)
    path = os.path.join(SCRATCH, "%s_%s.dat" % (prefix, tag))
    iq.tofile(path)
    rf = open_rawfile(pg, path, fs, T)
    rf.update_rawsnippet()
    Cf = int(rf.carr_fftpts)
    code, carr, cpc, win, idxn = [], [], [], [], []
    for k in range(len(ch["prn"])):
        cor = pg.correlator.Correlator(int(ch["prn"][k]))
        cc, cf, cp_compl = cor.vector_correlate_unfolded(
            rf, ch["rc"][k], ch["ri"][k], ch["fc"][k], ch["fi"][k], float(ch["cp"][k]), float(ch["cp_ref"][k]))
        code.append(np.asarray(cc)[S // 2 - 64: S // 2 + 65])
        carr.append(np.asarray(cf)[Cf // 2 - 256: Cf // 2 + 257])
        cpc.append(float(cp_compl))
    np.savez_compressed(os.path.join(HERE, "%s_%s.npz" % (prefix, tag)), iq=iq, fs=fs, T=T, S=S, N=rf.N, C=Cf,
                        first16=np.asarray(rf.rawsnippet)[:16], prn=ch["prn"], rc=ch["rc"], ri=ch["ri"],
                        fc=ch["fc"], fi=ch["fi"], cp=ch["cp"], cp_ref=ch["cp_ref"],
                        code=np.stack(code), carr=np.stack(carr), cp_compl=np.array(cpc), flip=np.asarray(flip))
    rf.close_rawfile()


def make_o12(pg):
    """O12: the same reference function at config H's sampling rate (25 Msps, 4 ms window): chips of 24 / 25 samples, one
    channel with a nav-bit sign change inside the window, one with the boundary inside but no change, one without a
    boundary -- the inputs of the chip-boundary stage-1 kernels."""
    ch = dpe.synth.random_channels(31, 3, prns=[7, 19, 30])
    ch["cp_ref"] = ch["cp"] - np.array([19, 19, 5], dtype=np.int32)   # edges 1 ms - code phase into the window (twice), none
    run_o3(pg, "highrate_4ms", 25e6, 0.004, 32, ch, 60.0, np.array([1, 0, 0], dtype=bool), prefix="o12")


def time_dp(pg, iters=3):
    """SURVEY 8d, CPU baseline (2): the PyGNSS DP path timed in the dev container -- per 20 ms window,
    dp_time_update_channels_unfolded (vector_correlate_unfolded x K, the BatchCorrScores twin) and
    dp_measurement_estimation_unfolded (the BatchCorrManifold twin incl. its grid generation).  Prints one JSON line;
    quoted in DESIGN.md section 5, never used by a test."""
    import json
    import time
    ho = dpe.handoff.read_handoff(os.path.join(REF, "demofiles", "handoff_params_usrp6.csv"))
    prns = [int(p) for p in ho["prn_list"]]
    ch_ho = dict(prn=ho["prn_list"], rc=ho["rc"], ri=ho["ri"], fc=ho["fc"], fi=ho["fi"], cp=ho["cp"],
                 cp_ref=ho["cp_timestamp"])
    fs, T = 2.5e6, 0.02
    S = int(round(fs * T))
    iq = np.concatenate([dpe.synth.gen_iq(21 + i, fs, S, ch_ho, amp=200.0, flip=np.zeros(len(prns), dtype=bool))
                         for i in range(iters + 1)])
    path = os.path.join(SCRATCH, "time_dp.dat")
    iq.tofile(path)
    rf = open_rawfile(pg, path, fs, T)
    rx = pg.receiver.Receiver(rf, mcount_max=iters + 4)
    rx.add_channels(prns)
    for k, p in enumerate(prns):
        rx.channels[p].ephemerides = Eph(ho, k)
    rx.ekf = pg.ekf.ExtendedKalmanFilter(np.asmatrix(ho["X_ECEF"]).T, T=T)
    rx.navguess = pg.receiver.NavigationGuesses()
    rx.rxTime = ho["rxTime"]
    rx.ekf.X_ECEF = np.matrix(ho["X_ECEF"]).T
    rx.rxTime_a = rx.rxTime - (rx.ekf.X_ECEF[3, 0] / 299792458.0)
    for k, p in enumerate(prns):
        c = rx.channels[p]
        c.rc[0], c.ri[0], c.fc[0], c.fi[0], c.cp[0] = ho["rc"][k], ho["ri"][k], ho["fc"][k], ho["fi"][k], float(ho["cp"][k])
    rf.seek_rawfile(rf.S_skip)
    t_bcs = t_bcm = 0.0
    for it in range(iters + 1):
        rf.update_rawsnippet()
        rx.dp_time_update_state()
        t0 = time.perf_counter()
        rx.dp_time_update_channels_unfolded()
        t1 = time.perf_counter()
        rx._mcount += 1
        rx.dp_measurement_estimation_unfolded()
        t2 = time.perf_counter()
        if it > 0:                                  # first window: warm-up
            t_bcs += t1 - t0
            t_bcm += t2 - t1
    rf.close_rawfile()
    G = 2 * 25 ** 4
    per = (t_bcs + t_bcm) / iters
    print(json.dumps({"windows": iters, "svs": len(prns), "bcs_twin_s_per_window": t_bcs / iters,
                      "bcm_twin_s_per_window": t_bcm / iters, "gridpoint_sv_per_s": G * len(prns) / per,
                      "x_realtime": 0.02 / per, "cores": 1, "numpy": np.__version__}))


def main():
    pg = import_pygnss()
    if "--time-dp" in sys.argv:
        time_dp(pg)
        return
    if "--only-o11" in sys.argv:
        make_o11(pg)
        return
    if "--only-o9" in sys.argv:
        make_o9(pg)
        return
    if "--only-o10" in sys.argv:
        make_o10(pg)
        return
    if "--only-o12" in sys.argv:
        make_o12(pg)
        return
    ho = dpe.handoff.read_handoff(os.path.join(REF, "demofiles", "handoff_params_usrp6.csv"))
    prns = [int(p) for p in ho["prn_list"]]
    K = len(prns)

    # ---- O1: C/A chips, PRN 1..37
    chips = np.stack([pg.correlator.Correlator(p).chips.astype(np.int8) for p in range(1, 38)])
    np.savez_compressed(os.path.join(HERE, "o1_ca_chips.npz"), chips=chips)

    # ---- O3: vector_correlate_unfolded on seeded synthetic windows (run_o3)
    ch_ho = dict(prn=ho["prn_list"], rc=ho["rc"], ri=ho["ri"], fc=ho["fc"], fi=ho["fi"], cp=ho["cp"],
                 cp_ref=ho["cp_timestamp"])
    flips = np.array([1, 0, 1, 1, 0, 0, 1, 0], dtype=bool)
    run_o3(pg, "handoff_20ms", 2.5e6, 0.02, 11, ch_ho, 200.0, flips)          # nav-bit edge inside window
    ch_b = dpe.synth.random_channels(5, 4)
    ch_b["cp_ref"] = ch_b["cp"] - np.array([0, 1, 2, 7], dtype=np.int32)  # next edge >= 13 ms away
    run_o3(pg, "short_5ms", 2.5e6, 0.005, 12, ch_b, 48.0, np.zeros(4, dtype=bool))  # no edge inside window

    # ---- O4: satellite clock correction + position/velocity at the handoff transmit times
    tx = ho["TOW"] + (ho["cp"] - ho["cp_timestamp"]) * 1e-3 + ho["rc"] / 1.023e6
    sat = np.zeros((K, 8))
    for k in range(K):
        e = Eph(ho, k)
        clkb, clkd = pg.satpos.satellite_clock_correction(e, tx[k])
        sat[k] = np.asarray(pg.satpos.locate_satellite(e, tx[k] - clkb, clkb, clkd)).ravel()
    np.savez_compressed(os.path.join(HERE, "o4_satpos.npz"), tx=tx, sat=sat, eph=ho["eph"], prn=ho["prn_list"])

    # ---- O5: ECEF->LLA, ENU rotation
    X = np.asmatrix(ho["X_ECEF"]).T
    lla = pg.utils.ECEF_to_LLA(X[0:3], in_degrees=False)
    _, Rm = pg.utils.ECEF_to_ENU(refState=X[0:3], curState=X[0:3])
    np.savez_compressed(os.path.join(HERE, "o5_frames.npz"), X_ECEF=ho["X_ECEF"], lat=lla["lat"][0],
                        lon=lla["lon"][0], alt=lla["alt"][0], R_ECEF2ENU=np.asarray(Rm))

    # ---- O6: spread grid
    ng = pg.receiver.NavigationGuesses()
    np.savez_compressed(os.path.join(HERE, "o6_spread_grid.npz"), dX=np.asarray(ng.dX), dT=np.asarray(ng.dT),
                        dXdot=np.asarray(ng.dXdot), dTdot=np.asarray(ng.dTdot))

    # ---- O7: one DP iteration on the handoff state with a geometry-consistent synthetic window
    fs, T = 2.5e6, 0.02
    S = int(round(fs * T))
    iq = dpe.synth.gen_iq(21, fs, S, ch_ho, amp=200.0, flip=flips)
    path = os.path.join(SCRATCH, "o7.dat")
    iq.tofile(path)
    rf = open_rawfile(pg, path, fs, T)
    rx = pg.receiver.Receiver(rf, mcount_max=8)
    rx.add_channels(prns)
    for k, p in enumerate(prns):
        rx.channels[p].ephemerides = Eph(ho, k)
    rx.ekf = pg.ekf.ExtendedKalmanFilter(np.asmatrix(ho["X_ECEF"]).T, T=T)
    rx.navguess = pg.receiver.NavigationGuesses()
    # what Receiver.load_cudarecv_handoff (receiver.py:129-178) sets, minus the file seek
    rx.rxTime = ho["rxTime"]
    rx.ekf.X_ECEF = np.matrix(ho["X_ECEF"]).T
    rx.rxTime_a = rx.rxTime - (rx.ekf.X_ECEF[3, 0] / 299792458.0)
    for k, p in enumerate(prns):
        c = rx.channels[p]
        c.rc[0], c.ri[0], c.fc[0], c.fi[0], c.cp[0] = ho["rc"][k], ho["ri"][k], ho["fc"][k], ho["fi"][k], float(ho["cp"][k])
    # Receiver.dp_track body (receiver.py:205-225), one iteration, with taps
    rf.seek_rawfile(rf.S_skip)
    rf.update_rawsnippet()
    rx.dp_time_update_state()
    rx.dp_time_update_channels_unfolded()
    rx._mcount += 1
    mc = rx._mcount
    Cf = int(rf.carr_fftpts)
    end = {n: np.array([getattr(rx.channels[p], n)[mc] for p in prns]) for n in ("rc", "ri", "fc", "fi", "cp")}
    code = np.stack([np.asarray(rx.channels[p].code_corr)[S // 2 - 64: S // 2 + 65] for p in prns])
    carr = np.stack([np.asarray(rx.channels[p].carr_fft)[Cf // 2 - 256: Cf // 2 + 257] for p in prns])
    X_ECEF = np.asarray(rx.ekf.X_ECEF).ravel().copy()
    gfv, gfp = rx.navguess.get_nav_guesses(rx.ekf.X_ECEF, rx.rxTime_a, ECEF_only=True)
    rx.dp_measurement_estimation_unfolded(gXk_grid=(gfv, gfp))
    pos_corr, vel_fft = np.asarray(rx.pos_corr).ravel(), np.asarray(rx.vel_fft).ravel()
    e = np.asarray(rx.dp_measurement_estimation_unfolded()).ravel()
    top_p = np.argsort(-pos_corr, kind="stable")[:32]
    top_v = np.argsort(-vel_fft, kind="stable")[:32]
    sel = [0, 1, 24, 25, 624, 625, 195312, 390624]
    np.savez_compressed(
        os.path.join(HERE, "o7_dp_iteration.npz"), iq=iq, fs=fs, T=T, S=S, C=Cf, prn=ho["prn_list"],
        X_ECEF=X_ECEF, rxTime=rx.rxTime, rxTime_a=rx.rxTime_a, end_rc=end["rc"], end_ri=end["ri"],
        end_fc=end["fc"], end_fi=end["fi"], end_cp=end["cp"], code=code, carr=carr,
        argmax_pos=int(np.argmax(pos_corr)), argmax_vel=int(np.argmax(vel_fft)), e=e,
        pos_every97=pos_corr[::97], vel_every97=vel_fft[::97], top_pos_idx=top_p, top_pos=pos_corr[top_p],
        top_vel_idx=top_v, top_vel=vel_fft[top_v], gX_sel=np.asarray(gfv)[:, sel], sel=np.array(sel))
    rf.close_rawfile()
    # ---- O8: cold-start coarse acquisition (correlator.py:53-103) on a 10 ms synthetic window
    fs, T = 2.5e6, 0.01
    S = int(round(fs * T))
    ch8 = dpe.synth.random_channels(8, 4, prns=[2, 12, 19, 28])
    ch8["fi"] = np.array([1234.0, -2750.0, 310.0, 4490.0])          # off the 100 Hz search raster
    ch8["fc"] = 1.023e6 * (1.0 + ch8["fi"] / 1.57542e9)
    ch8["cp_ref"] = ch8["cp"].copy()                                # no nav-bit edge inside 10 ms
    iq = dpe.synth.gen_iq(31, fs, S, ch8, amp=np.array([120.0, 90.0, 60.0, 150.0]), flip=np.zeros(4, dtype=bool))
    path = os.path.join(SCRATCH, "o8.dat")
    iq.tofile(path)
    rf = open_rawfile(pg, path, fs, T)
    rf.update_rawsnippet()

    class Py2Int(int):           # HARNESS: Python-2 "int / int" floors (correlator.py:78 reshapes by S/N)
        def __truediv__(self, other):
            return int(self) // other
    rf.S = Py2Int(rf.S)
    cases = []
    for prn in (2, 12, 19, 28, 5, 30):                              # four present, two absent
        for coherent, mat in ((True, pg.correlator.DOPPLER_SEARCH_MATRIX_COHERENT),
                              (False, pg.correlator.DOPPLER_SEARCH_MATRIX_NONCOHERENT)):
            cor = pg.correlator.Correlator(prn)
            m, found, rc, fc, fi, cppr, cppm = cor.coarse_acquisition(rf, mat, coherent, False)
            a = np.abs(np.asarray(m))
            mp = a.max(0)
            ci = int(mp.argmax())
            di = int(a[:, ci].argmax())
            nb = a[max(di - 8, 0):di + 9, :][:, np.arange(ci - 8, ci + 9) % a.shape[1]]
            cases.append(dict(prn=prn, coherent=coherent, found=bool(found), rc=float(rc), fc=float(fc), fi=float(fi),
                              cppr=float(cppr), cppm=float(cppm), ci=ci, di=di, nb=nb, shape=a.shape,
                              row_max=a.max(1), col_every25=mp[::25]))
    np.savez_compressed(os.path.join(HERE, "o8_acquisition.npz"), iq=iq, fs=fs, T=T, S=S, N=rf.N,
                        bins_coh=np.asarray(pg.correlator.DOPPLER_SEARCH_MATRIX_COHERENT).ravel(),
                        bins_non=np.asarray(pg.correlator.DOPPLER_SEARCH_MATRIX_NONCOHERENT).ravel(),
                        truth_prn=ch8["prn"], truth_rc=ch8["rc"], truth_fi=ch8["fi"],
                        **{"c%d_%s" % (i, k): np.asarray(v) for i, c in enumerate(cases) for k, v in c.items()},
                        ncases=len(cases))
    rf.close_rawfile()
    make_o9(pg)
    make_o10(pg)
    make_o11(pg)
    make_o12(pg)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
