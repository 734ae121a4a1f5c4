"""RINEX 2.x GPS navigation-message reader and the reference's ephemeris choice.

The reference's DPInit reads a RINEX nav file next to the handoff CSV (cudarecv/modules/src/dpinit.cpp:130-144;
parser adapted from RTKLIB in cudarecv/utils/src/rinexparse.cpp), groups the records into sets by integer toe
(ReadRinexBody :186-222) and cuChanMgr picks, per PRN, the set whose toe is closest to the transmit time
(CHM_ComputeSatStates, cuchanmgr.cu:269-299: first valid set, replaced only by a strictly closer one; seconds of
week only, no week number).  PyGNSS twin: libgnss/rinex.py (first record of a PRN).

Record layout (RINEX 2.10, "GPS NAV MESSAGE FILE - DATA RECORD DESCRIPTION"): 8 lines per record; line 1 =
PRN I2, epoch (toc) yy mm dd hh mm ss.s, then af0 af1 af2 as D19.12; lines 2-8 = 3X, 4 x D19.12."""
import datetime

import numpy as np

from .handoff import EPH_FIELDS

# index into the 31 data fields of a record (af0, af1, af2, then 7 lines x 4), rinexparse.cpp:324-345
_FIELD_INDEX = {"sqrt_A": 10, "e": 8, "i_0": 15, "OMEGA_0": 13, "omega": 17, "M_0": 6, "delta_n": 5, "OMEGADOT": 18,
                "IDOT": 19, "C_rc": 16, "C_rs": 4, "C_uc": 7, "C_us": 9, "C_ic": 12, "C_is": 14, "t_oe": 11,
                "a_f0": 0, "a_f1": 1, "a_f2": 2, "T_GD": 25}
_GPS_EPOCH = datetime.datetime(1980, 1, 6)


def _num(s):
    s = s.strip().replace("D", "E").replace("d", "E")
    return float(s) if s else 0.0


def gps_seconds_of_week(year, month, day, hour, minute, second):
    """(week, seconds of week) of a GPS-time calendar epoch (time2gpst)."""
    whole = int(second)
    dt = datetime.datetime(year, month, day, hour, minute, whole) - _GPS_EPOCH
    total = dt.days * 86400 + dt.seconds
    return total // 604800, float(total % 604800) + (second - whole)


def read_rinex_nav(path):
    """-> dict(prn [n] int, week [n] int, tocs [n], data [n,31], eph [n,21] in handoff.EPH_FIELDS order), file order."""
    with open(path, "r") as f:
        lines = f.read().splitlines()
    i = 0
    version = 2.10
    while i < len(lines):
        label = lines[i][60:] if len(lines[i]) > 60 else ""
        if "RINEX VERSION / TYPE" in label:
            version = _num(lines[i][0:9])
            if lines[i][20] != "N":
                raise ValueError("not a RINEX navigation file: type %r" % lines[i][20])
        i += 1
        if "END OF HEADER" in label:
            break
    if version >= 3.0:
        raise ValueError("RINEX version %.2f unsupported (the reference parser handles < 3.0)" % version)
    prn, week, tocs, data = [], [], [], []
    while i + 7 < len(lines):
        l0 = lines[i]
        if not l0.strip():
            i += 1
            continue
        p = int(l0[0:2])
        yy, mo, dd, hh, mi = (int(l0[k:k + 2]) for k in (3, 6, 9, 12, 15))
        ss = _num(l0[17:22])
        year = yy + (2000 if yy < 80 else 1900)
        wk, sow = gps_seconds_of_week(year, mo, dd, hh, mi, ss)
        vals = [_num(l0[22 + 19 * j:22 + 19 * (j + 1)]) for j in range(3)]
        for r in range(1, 8):
            ln = lines[i + r].ljust(80)
            vals += [_num(ln[3 + 19 * j:3 + 19 * (j + 1)]) for j in range(4)]
        prn.append(p); week.append(wk); tocs.append(sow); data.append(vals)
        i += 8
    data = np.array(data, dtype=np.float64).reshape(-1, 31)
    tocs = np.array(tocs, dtype=np.float64)
    eph = np.zeros((data.shape[0], len(EPH_FIELDS)))
    for j, name in enumerate(EPH_FIELDS):
        eph[:, j] = tocs if name == "t_oc" else data[:, _FIELD_INDEX[name]]
    return dict(prn=np.array(prn, dtype=np.int64), week=np.array(week, dtype=np.int64), tocs=tocs, data=data, eph=eph)


def select_ephemerides(nav, prn_list, tx_time):
    """The reference's choice (cuchanmgr.cu:269-299): sets are keyed by int(toe) in order of first appearance, a later
    record of the same (toe, PRN) replaces the earlier one; per PRN the first set holding it wins unless a later set's
    toe is STRICTLY closer to tx_time (scalar or per-PRN array, seconds of week).  -> eph [K, 21]."""
    tx = np.broadcast_to(np.asarray(tx_time, dtype=np.float64), (len(prn_list),))
    toe_idx = EPH_FIELDS.index("t_oe")
    sets, order = {}, []                       # int(toe) -> {prn: record index}
    for r in range(nav["prn"].size):
        key = int(nav["eph"][r, toe_idx])
        if key not in sets:
            sets[key] = {}
            order.append(key)
        sets[key][int(nav["prn"][r])] = r
    out = np.zeros((len(prn_list), len(EPH_FIELDS)))
    for k, p in enumerate(prn_list):
        best = None
        for key in order:
            if int(p) not in sets[key]:
                continue
            if best is None or abs(key - tx[k]) < abs(best - tx[k]):
                best = key
        if best is None:
            raise ValueError("no ephemeris for PRN %d in the RINEX data" % int(p))
        out[k] = nav["eph"][sets[best][int(p)]]
    return out
