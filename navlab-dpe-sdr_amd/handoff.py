"""Reader for the PyGNSS->CUDARecv handoff CSV (demofiles/handoff_params_usrp6.csv).

Keys and meaning follow DPInit::ParseField (cudarecv/modules/src/dpinit.cpp:247-400) and the
writer pygnss/pythonreceiver/receiver.py:804-875.  Rows: key,v0,v1,...; per-PRN rows are in
prn_list order.  Ephemeris rows (T_GD ... C_is) are the broadcast ephemeris PyGNSS decoded.
"""
import numpy as np

EPH_FIELDS = ["sqrt_A", "e", "i_0", "OMEGA_0", "omega", "M_0", "delta_n", "OMEGADOT", "IDOT",
              "C_rc", "C_rs", "C_uc", "C_us", "C_ic", "C_is", "t_oe", "t_oc",
              "a_f0", "a_f1", "a_f2", "T_GD"]


def read_handoff(path, rinex_path=None):
    """Handoff CSV -> dict.  rinex_path: take the ephemerides from a RINEX nav file instead of the handoff rows, chosen
    as the reference does (DPInit + cuChanMgr, see rinex.py)."""
    rows = {}
    with open(path, "r") as f:
        for line in f:
            parts = line.strip().split(",")
            if len(parts) < 2:
                continue
            rows[parts[0]] = parts[1:]
    out = {
        "rxTime": float(rows["rxTime"][0]),
        "rxTime_a": float(rows["rxTime_a"][0]),
        "X_ECEF": np.array([float(v) for v in rows["X_ECEF"]]),
        "bytes_read": int(rows["bytes_read"][0]),
        "prn_list": np.array([int(v) for v in rows["prn_list"]], dtype=np.int32),
    }
    for k in ("rc", "ri", "fc", "fi"):
        out[k] = np.array([float(v) for v in rows[k]])
    out["cp"] = np.array([int(float(v)) for v in rows["cp"]], dtype=np.int32)
    out["cp_timestamp"] = np.array([int(float(v)) for v in rows["cp_timestamp"]], dtype=np.int32)
    out["TOW"] = np.array([int(float(v)) for v in rows["TOW"]], dtype=np.int32)
    K = out["prn_list"].size
    if rinex_path is not None:
        from . import rinex
        out["eph"] = rinex.select_ephemerides(rinex.read_rinex_nav(rinex_path), out["prn_list"], out["rxTime"])
        return out
    eph = np.zeros((K, len(EPH_FIELDS)))
    for j, name in enumerate(EPH_FIELDS):
        eph[:, j] = [float(v) for v in rows[name]]
    out["eph"] = eph
    return out
