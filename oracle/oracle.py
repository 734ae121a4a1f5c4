"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product package).

ctypes binding of oracle/libdpe_oracle.so (plain-C fp64 restatement, CUDARecv semantics)
plus a numpy FFT-based full-length restatement of BatchCorrScores::Update that follows the
reference step by step (cudarecv/modules/src/batchcorrscores.cu:1043-1180, twin:
pygnss/pythonreceiver/scalar/correlator.py:367-465).

Pinned against tests/golden/*.npz (generated from the reference's Python twin by
tests/golden/make_golden.py); see oracle/dpe_oracle.h for the parity-pin statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EPH_FIELDS = ["sqrt_A", "e", "i_0", "OMEGA_0", "omega", "M_0", "delta_n", "OMEGADOT", "IDOT",
              "C_rc", "C_rs", "C_uc", "C_us", "C_ic", "C_is", "t_oe", "t_oc",
              "a_f0", "a_f1", "a_f2", "T_GD"]
EPH_N = len(EPH_FIELDS)

# constants: cudarecv/utils/inc/consthelper.h:5-27
CONST_C = 299792458.0
CONST_PI = 3.1415926535898
F_L1 = 1.57542e9
F_CA = 1.023e6
L_CA = 1023
T_CA = 0.001
OEDOT = 7.2921151467e-5


def build(force=False):
    so = os.path.join(_HERE, "libdpe_oracle.so")
    src = os.path.join(_HERE, "dpe_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.dpo_time_idx.restype = C.c_double
        _LIB.dpo_time_idx.argtypes = [C.c_int64, C.c_double]
        _LIB.dpo_argmax_first.restype = C.c_int64
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _d(a):
    return _p(a, C.c_double)


def _i(a):
    return _p(a, C.c_int)


def ca_code(prn):
    out = np.zeros(1023, dtype=np.int8)
    lib().dpo_gen_ca_code(C.c_int(int(prn)), _p(out, C.c_int8))
    return out


def nav_bit_boundary(cp_ela, cp_ref, rc, fc, fs):
    f = lib().dpo_nav_bit_boundary
    f.restype = C.c_int
    return f(C.c_int(int(cp_ela)), C.c_int(int(cp_ref)), C.c_double(rc), C.c_double(fc), C.c_double(fs))


def carr_fft_len(S):
    """batchcorrscores.cu:761 carrSTot = 8 * roundUpToNextPowerOfTwo(S) (auxil.cpp:98-109)."""
    p = 1
    while p < S:
        p <<= 1
    return 8 * p


def bcs_sv(iq, fs, prn, rc, ri, fc, fi, cp_ela, cp_ref, lag_lo, lag_hi, bin_lo, bin_hi, C_fft=None):
    """One SV of BatchCorrScores (direct-sum C oracle).  iq: int16[2*S] interleaved."""
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    S = iq.size // 2
    if C_fft is None:
        C_fft = carr_fft_len(S)
    chips = ca_code(prn)
    code = np.zeros(2 * (lag_hi - lag_lo + 1))
    carr = np.zeros(2 * (bin_hi - bin_lo + 1))
    info = np.zeros(2, dtype=np.int32)
    mean = np.zeros(2)
    rcode = lib().dpo_bcs_sv(_p(iq, C.c_int16), C.c_int(S), C.c_double(fs), _p(chips, C.c_int8),
                             C.c_double(rc), C.c_double(ri), C.c_double(fc), C.c_double(fi),
                             C.c_int(int(cp_ela)), C.c_int(int(cp_ref)),
                             C.c_int(lag_lo), C.c_int(lag_hi), C.c_int64(C_fft),
                             C.c_int(bin_lo), C.c_int(bin_hi), _d(code), _d(carr), _i(info), _d(mean))
    assert rcode == 0
    return (code.view(np.complex128), carr.view(np.complex128),
            dict(idx_next=int(info[0]), no_flip_larger=bool(info[1]), mean=complex(mean[0], mean[1])))


def bcs_sv_fft(iq, fs, prn, rc, ri, fc, fi, cp_ela, cp_ref, C_fft=None):
    """Full-length FFT restatement of one SV of BatchCorrScores::Update (numpy).

    Returns (codeScores[S], carrScores[C], info), both fft-shifted exactly like
    codeCorrOut_d / carrfftOut_d (batchcorrscores.cu:1153,1180)."""
    iq = np.asarray(iq, dtype=np.int16)
    S = iq.size // 2
    if C_fft is None:
        C_fft = carr_fft_len(S)
    raw = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)        # :216
    t = np.round(np.arange(S) / fs * 1.0e9) / 1.0e9                            # :191-193
    mean = raw.sum() / np.float64(np.float32(S))                                # :1065
    wipe = np.conj(np.exp(1j * (2 * CONST_PI * (fi * t + ri))))                 # :294-300
    chips = ca_code(prn).astype(np.float64)
    r = chips[np.mod(np.floor(t * fc + rc).astype(np.int64), L_CA)]             # :347-349
    idx_next = nav_bit_boundary(cp_ela, cp_ref, rc, fc, fs)                     # :247-253
    has_flip = 0 < idx_next < S
    rf = np.zeros(S)
    if has_flip:                                                                # :352-367
        rf = r.copy()
        rf[idx_next:] = -rf[idx_next:]
    b = raw * wipe                                                              # :1113
    bf = np.fft.fft(b)                                                          # :1120
    c_no = np.fft.ifft(np.conj(np.fft.fft(r)) * bf)                             # :1099-1144
    c_fl = np.fft.ifft(np.conj(np.fft.fft(rf)) * bf)
    no_flip_larger = (not has_flip) or (abs(c_no[0]) > abs(c_fl[0]))            # :512
    code = np.fft.fftshift(c_no if no_flip_larger else c_fl)                    # :1153
    rch = r if no_flip_larger else rf
    base = (b - mean * wipe) * rch                                              # :480,:440-448
    carr = np.fft.fftshift(np.fft.fft(base, C_fft))                             # :1179-1180
    return code, carr, dict(idx_next=idx_next, no_flip_larger=bool(no_flip_larger), mean=complex(mean))


def bcm_pos(sat, code_win, win_lo, center, grid, R, fc, cp_ref_tow, cp_ela_end, cp_ref, rc_end,
            rx_time, fs, num_samps, lpower=1, extended=False):
    """BCM_PosMeasML.  sat: K x 8 mid-time states; code_win: K x winLen complex.
    extended=True evaluates the index in long double (measures the reference's fp64 noise)."""
    if extended:
        lpower = -int(lpower)
    sat = np.ascontiguousarray(sat, dtype=np.float64)
    K = sat.shape[0]
    cw = np.ascontiguousarray(code_win, dtype=np.complex128)
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    G = grid.shape[0]
    scores = np.zeros(G)
    oob = C.c_int64(0)
    ii = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    dd = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    a_tow, a_ela, a_ref = ii(cp_ref_tow), ii(cp_ela_end), ii(cp_ref)
    a_fc, a_rc, a_c, a_R = dd(fc), dd(rc_end), dd(center), dd(R)
    lib().dpo_bcm_pos(_d(sat), _d(cw.view(np.float64)), C.c_int(int(win_lo)), C.c_int(cw.shape[1]),
                      _d(a_c), _d(grid), C.c_int64(G), _d(a_R), _d(a_fc), _i(a_tow), _i(a_ela),
                      _i(a_ref), _d(a_rc), C.c_double(rx_time), C.c_int(K), C.c_double(fs),
                      C.c_int(int(num_samps)), C.c_int(int(lpower)), _d(scores), C.byref(oob))
    return scores, oob.value


def bcm_pos_quirks():
    """Indices of the grid points at which the last faithful bcm_pos call took the reference's double-counting branch
    (floor(idx) and floor(idx+1) two apart; see dpe_oracle.c)."""
    buf = np.zeros(256, dtype=np.int64)
    lib().dpo_bcm_pos_quirks.restype = C.c_int64
    n = lib().dpo_bcm_pos_quirks(_p(buf, C.c_int64), C.c_int64(256))
    return buf[:min(n, 256)].copy()


def bcm_vel(sat, carr_win, win_lo, center, grid, R, fi, rx_time, fs, num_fft, doppler_sign=1, lpower=1):
    sat = np.ascontiguousarray(sat, dtype=np.float64)
    K = sat.shape[0]
    cw = np.ascontiguousarray(carr_win, dtype=np.complex128)
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    G = grid.shape[0]
    scores = np.zeros(G)
    oob = C.c_int64(0)
    dd = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    a_fi, a_c, a_R = dd(fi), dd(center), dd(R)
    lib().dpo_bcm_vel(_d(sat), _d(cw.view(np.float64)), C.c_int64(int(win_lo)), C.c_int(cw.shape[1]),
                      _d(a_c), _d(grid), C.c_int64(G), _d(a_R), _d(a_fi), C.c_double(rx_time),
                      C.c_int(K), C.c_double(fs), C.c_int64(int(num_fft)), C.c_int(int(doppler_sign)),
                      C.c_int(int(lpower)), _d(scores), C.byref(oob))
    return scores, oob.value


def argmax_first(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    return int(lib().dpo_argmax_first(_d(v), C.c_int64(v.size)))


def make_meas(pos_idx, vel_idx, center, pos_grid, vel_grid, R):
    z = np.zeros(8)
    Rv = np.zeros(64)
    dd = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    a_c, a_pg, a_vg, a_R = dd(center), dd(pos_grid), dd(vel_grid), dd(R)
    lib().dpo_make_meas(C.c_int64(int(pos_idx)), C.c_int64(int(vel_idx)), _d(a_c), _d(a_pg), _d(a_vg),
                        _d(a_R), _d(z), _d(Rv))
    return z, Rv.reshape(8, 8)


def init_grid(grid_type, dims, spacing):
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    spacing = np.ascontiguousarray(spacing, dtype=np.float64)
    G = int(np.prod(dims.astype(np.int64)))
    grid = np.zeros((G, 4))
    tg = np.zeros(int(dims[3]))
    lib().dpo_init_grid(C.c_int(int(grid_type)), _i(dims), _d(spacing), _d(grid), _d(tg))
    return grid, tg


def sat_pos(eph, tx_time):
    eph = np.ascontiguousarray(eph, dtype=np.float64)
    st = np.zeros(8)
    f = lib().dpo_sat_pos
    f.restype = C.c_int
    rc = f(_d(eph), C.c_double(tx_time), _d(st))
    return st, rc


def ecef2ll(p):
    p = np.ascontiguousarray(p[:3], dtype=np.float64)
    ll = np.zeros(2)
    lib().dpo_ecef2ll(_d(p), _d(ll))
    return ll


def enu2ecef(ll):
    ll = np.ascontiguousarray(ll, dtype=np.float64)
    R = np.zeros(9)
    lib().dpo_enu2ecef(_d(ll), _d(R))
    return R


class _ChanT(C.Structure):
    _fields_ = [("K", C.c_int)] + \
               [(n, C.POINTER(C.c_double)) for n in ("rcStart", "rcEnd", "riStart", "riEnd", "fc", "fi", "txTime")] + \
               [(n, C.POINTER(C.c_int)) for n in ("cpElaStart", "cpElaEnd", "cpRef", "cpRefTOW")] + \
               [("satStates", C.POINTER(C.c_double)), ("eph", C.POINTER(C.c_double)), ("dopplerSign", C.c_int)]


class ChanMgr:
    """cuChanMgr restatement (cuchanmgr.cu:1004-1268): Start() then Update() per window."""

    def __init__(self, prns, rc, ri, fc, fi, cp, cp_ref, cp_ref_tow, eph, rx_time, T, doppler_sign=1):
        K = len(prns)
        self.K, self.prns, self.T = K, list(prns), float(np.round(T * 1e6) / 1e6)
        d = lambda a: np.array(a, dtype=np.float64).copy()
        i = lambda a: np.array(a, dtype=np.int32).copy()
        self.rcStart, self.rcEnd = np.zeros(K), d(rc)
        self.riStart, self.riEnd = np.zeros(K), d(ri)
        self.fc, self.fi, self.txTime = d(fc), d(fi), np.zeros(K)
        self.cpElaStart, self.cpElaEnd = np.zeros(K, dtype=np.int32), i(cp)
        self.cpRef, self.cpRefTOW = i(cp_ref), i(cp_ref_tow)
        self.satStates = np.zeros((K, 8))
        self.eph = np.ascontiguousarray(eph, dtype=np.float64)
        self.rxTime = float(rx_time)
        self.dopplerSign = int(doppler_sign)
        self._s = _ChanT(K, _d(self.rcStart), _d(self.rcEnd), _d(self.riStart), _d(self.riEnd),
                         _d(self.fc), _d(self.fi), _d(self.txTime), _i(self.cpElaStart),
                         _i(self.cpElaEnd), _i(self.cpRef), _i(self.cpRefTOW), _d(self.satStates),
                         _d(self.eph), self.dopplerSign)

    def start(self, x_k1k1, x_kk1, time_grid):
        """cuchanmgr.cu:1100-1132."""
        x1 = np.ascontiguousarray(x_k1k1, dtype=np.float64)
        lib().dpo_chm_compute_sat_states(C.byref(self._s))
        lib().dpo_chm_time_update(C.byref(self._s), _d(x1), C.c_double(self.rxTime), C.c_double(self.T))
        self.rxTime += self.T
        return self.grid_prep(x_kk1, time_grid)

    def update(self, x_k1k1, x_kk1, time_grid):
        """cuchanmgr.cu:1237-1264."""
        x1 = np.ascontiguousarray(x_k1k1, dtype=np.float64)
        lib().dpo_chm_propagate(C.byref(self._s), _d(x1), C.c_double(self.rxTime), C.c_double(self.T))
        self.rxTime += self.T
        return self.grid_prep(x_kk1, time_grid)

    def grid_prep(self, x_kk1, time_grid):
        x = np.ascontiguousarray(x_kk1, dtype=np.float64)
        tg = np.ascontiguousarray(time_grid, dtype=np.float64)
        batch = np.zeros((self.K, tg.size, 8))
        R = np.zeros(9)
        lib().dpo_chm_grid_prep(C.c_double(self.rxTime), _d(self.txTime), _d(x), _d(self.satStates),
                                C.c_int(self.K), _d(tg), C.c_int(tg.size), _d(batch), _d(R))
        self.batchSatStates, self.enu2ecef = batch, R
        return batch, R


# ---------------------------------------------------------------------------------------------
# Cold-start acquisition.  Exists only in the reference's Python twin:
# pygnss/pythonreceiver/scalar/correlator.py:53-103 (coarse_acquisition), :13-14 (search grids).
def acq_bins(coherent=True):
    """DOPPLER_SEARCH_MATRIX_COHERENT (125 x 100 Hz) / _NONCOHERENT (25 x 500 Hz), correlator.py:13-14."""
    B, d = (125, 100.0) if coherent else (25, 500.0)
    return np.arange((1 - B) / 2, (B - 1) / 2 + 1) * d


def trim_mean(arr, percent):
    """correlator.py:546-564: mean of the values strictly between the percent/2 and 100-percent/2
    percentiles (scipy scoreatpercentile = linear interpolation)."""
    lo = np.percentile(arr, percent / 2.0)
    hi = np.percentile(arr, 100.0 - percent / 2.0)
    sel = arr[(arr > lo) & (arr < hi)]
    return sel.mean()


def coarse_acquisition(iq, fs, prn, bins, coherent=True, mode=None, doppler_sign=1.0, wrap_mask=False):
    """numpy restatement of Correlator.coarse_acquisition (correlator.py:53-103).

    mode None -> the reference's semantics (coherent flag as given).  mode 'textbook' -> the
    BASELINE.json config-5 wording (1 ms coherent x N non-coherent): NOT a reference algorithm,
    parity unpinned, provided for comparison only.

    The reference masks +-ceil(fs/F_CA) delays about the peak with a plain index array (correlator.py:96-99): negative
    indices wrap, but a peak within that distance of the LAST delay indexes past the end and the reference raises
    IndexError -- as does this restatement.  wrap_mask=True wraps both ends instead (what the HIP path does)."""
    iq = np.asarray(iq, dtype=np.int16)
    S = iq.size // 2
    N = int(round(S / fs / T_CA))
    raw = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    t = np.arange(S) / fs                                                     # rawfile.py:164-165
    code_idc = t * F_CA
    chips = ca_code(prn).astype(np.float64)
    rep = chips[np.mod(np.floor(code_idc), L_CA).astype(np.int64)]             # :66
    bins = np.asarray(bins, dtype=np.float64)
    M = S // N
    if mode == "textbook":
        R1 = np.conj(np.fft.fft(rep[:M]))
        res = np.zeros((bins.size, M))
        for b, f in enumerate(bins):
            x = (raw * np.exp(-1j * (2 * CONST_PI * f * t))).reshape(N, M)
            res[b] = np.abs(np.fft.ifft(np.fft.fft(x, axis=1) * R1[None, :], axis=1)).sum(0)
        res_abs = res
    else:
        Rc = np.conj(np.fft.fft(rep))                                          # :67
        res = np.zeros((bins.size, S), dtype=np.complex128)
        for b, f in enumerate(bins):                                           # :73-75
            res[b] = np.fft.ifft(np.fft.fft(raw * np.exp(-1j * (2 * CONST_PI * f * t))) * Rc)
        if N != 1:                                                             # :77-82
            tmp = res.reshape(bins.size, N, M)
            res = tmp.sum(1) if coherent else np.abs(tmp).sum(1)
        res_abs = np.abs(res)
    max_percode = res_abs.max(0)                                               # :87-89
    max_code_idx = int(max_percode.argmax())
    max_dopp_idx = int(res_abs[:, max_code_idx].argmax())
    rc = L_CA - code_idc[max_code_idx]
    fi = bins[max_dopp_idx]
    fc = F_CA + (doppler_sign * F_CA / F_L1) * fi
    peak = max_percode[max_code_idx]
    mask_S = int(np.ceil(fs / F_CA))
    mp = max_percode.copy()
    mask_idx = np.arange(-mask_S, mask_S + 1) + max_code_idx
    mp[mask_idx % mp.size if wrap_mask else mask_idx] = 0                      # negative indices wrap, as in numpy
    cppr = peak / mp.max()
    cppm = peak / trim_mean(mp, 10)
    return dict(surface=res_abs, max_code_idx=max_code_idx, max_dopp_idx=max_dopp_idx, rc=rc, fi=fi, fc=fc,
                cppr=cppr, cppm=cppm, found=bool(cppm > 2.0), max_percode=max_percode)


def fine_fft_len(S):
    """rawfile.carr_fftpts (rawfile.py:173): 8 * (1 << S.bit_length()) -- NOT BatchCorrScores' 8 * 2^ceil(log2 S)
    (they differ when S is a power of two)."""
    return 8 * (1 << int(S).bit_length())


def fine_frequency_acquisition(iq, fs, prn, rc, fc, bins, doppler_sign=1.0):
    """numpy restatement of Correlator.fine_frequency_acquisition (correlator.py:105-133): code wipe-off with
    the coarse (rc, fc), zero-padded FFT, peak inside [min(bins), max(bins)] -> ri (cycles), fi, fc."""
    iq = np.asarray(iq, dtype=np.int16)
    S = iq.size // 2
    raw = iq[0::2].astype(np.float64) + 1j * iq[1::2].astype(np.float64)
    t = np.arange(S) / fs
    chips = ca_code(prn).astype(np.float64)
    rep = chips[np.mod(np.floor(t * fc + rc), L_CA).astype(np.int64)]          # :114-115
    carr = (raw - np.mean(raw)) * rep                                          # :118
    C = fine_fft_len(S)
    X = np.fft.fftshift(np.fft.fft(carr, C))                                   # :121
    fidc = np.fft.fftshift(np.fft.fftfreq(n=C, d=1.0 / fs))                    # rawfile.py:174
    bins = np.asarray(bins, dtype=np.float64)
    X[fidc < bins.min()] = 0.0                                                 # :124-125
    X[fidc > bins.max()] = 0.0
    idx = int(np.abs(X).argmax())
    ri = np.angle(X[idx]) / (2.0 * CONST_PI)
    fi = fidc[idx]
    return dict(rc=rc, ri=ri, fc=F_CA + (doppler_sign * F_CA / F_L1) * fi, fi=fi, max_carr_idx=idx, peak=X[idx], C=C)


def search_signal(iq, fs, prn, bins=None, coherent=True, doppler_sign=1.0):
    """Correlator.search_signal (correlator.py:38-51): coarse, then fine frequency."""
    bins = acq_bins(coherent) if bins is None else bins
    c = coarse_acquisition(iq, fs, prn, bins, coherent=coherent, doppler_sign=doppler_sign)
    f = fine_frequency_acquisition(iq, fs, prn, c["rc"], c["fc"], bins, doppler_sign)
    return dict(found=c["found"], rc=f["rc"], ri=f["ri"], fc=f["fc"], fi=f["fi"], cppr=c["cppr"], cppm=c["cppm"],
                max_carr_idx=f["max_carr_idx"], max_code_idx=c["max_code_idx"], max_dopp_idx=c["max_dopp_idx"])


def scalar_acquisition(iq_two_windows, fs, prn_list, doppler_sign=1.0):
    """Receiver.scalar_acquisition (receiver.py:452-520): search two consecutive windows, keep the one with the
    larger cppm; a second-window hit is propagated back by one window (rc - fc T, ri - fi T).  Returns
    (per_window [2][P] dicts, final [P] (rc, ri, fc, fi)) -- parameters at the start of the FIRST window."""
    iq = np.asarray(iq_two_windows, dtype=np.int16)
    S = iq.size // 4
    T = S / fs
    wins = [[search_signal(iq[2 * S * w:2 * S * (w + 1)], fs, p, doppler_sign=doppler_sign) for p in prn_list] for w in range(2)]
    final = []
    for a, b in zip(*wins):
        if b["cppm"] > a["cppm"]:                                              # :493-498
            final.append((np.mod(b["rc"] - b["fc"] * T, L_CA), np.mod(b["ri"] - b["fi"] * T, 1.0), b["fc"], b["fi"]))
        else:                                                                  # :507
            final.append((a["rc"], a["ri"], a["fc"], a["fi"]))
    return wins, np.array(final)


class Ekf8:
    """numpy restatement of the 8-state Kalman filter (cuekf.cu:626-742 / ekf.py:58-177, the `_m5` methods).
    couple_velocity: F = I + T on [i][i+4] (CUDARecv, cuekf.cu:111-143) or F = I (PyGNSS as shipped, ekf.py:47)."""
    Q_CLOCK_DRIFT = ((2.5e-10) * CONST_C) ** 2.0                                 # ekf.py:69, cuekf.h:28

    def __init__(self, x0, P0=None, T=0.02, couple_velocity=True):
        self.F = np.eye(8)
        if couple_velocity:
            for j in range(4):
                self.F[j, j + 4] = T
        self.H = np.eye(8)
        self.x = np.array(x0, dtype=np.float64).reshape(8)        # x_k|k (after update) or x_k|k-1 (after predict)
        self.P = np.eye(8) if P0 is None else np.array(P0, dtype=np.float64).reshape(8, 8)
        self.Q = np.eye(8)
        self.K = np.eye(8)
        self.lpf = [0.0] * 20                                      # filters.RunningAverageFilter(20)

    def predict(self):                                             # _time_update_m5, ekf.py:171-178
        self.x = self.F @ self.x
        v = float(np.linalg.norm(self.x[4:7]))
        self.lpf.pop(0); self.lpf.append(v)
        vbar = sum(self.lpf) / 20.0
        q = 1.0 + 250.0 / min(max(vbar ** 2.0, 50.0), 125.0)       # :62
        Q = np.zeros((8, 8))
        Q[4, 4] = Q[5, 5] = Q[6, 6] = q
        Q[7, 7] = self.Q_CLOCK_DRIFT
        self.Q = self.F @ Q @ self.F.T                             # :70
        self.P = self.F @ self.P @ self.F.T + self.Q               # :77
        return self.x

    def update(self, z, R=None):                                   # _measurement_update_m5, ekf.py:160-169
        R = np.eye(8) if R is None else np.asarray(R, dtype=np.float64)
        e = np.asarray(z, dtype=np.float64).reshape(8) - self.H @ self.x
        S_inv = np.linalg.inv(self.H @ self.P @ self.H.T + R)      # :82
        self.K = self.P @ self.H.T @ S_inv                         # :84
        self.x = self.x + self.K @ e
        self.P = (np.eye(8) - self.K @ self.H) @ self.P            # :117
        return self.x
