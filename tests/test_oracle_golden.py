"""CPU: pin the oracle (oracle/) against the fixtures generated from the reference's Python
twin (tests/golden/make_golden.py).  These are the parity pins SURVEY.md 8c lists (O1-O7)."""
import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe


def test_o1_ca_chips(golden, oracle):
    g = golden("o1_ca_chips")["chips"]
    assert g.shape == (37, 1023)
    for prn in range(1, 38):
        assert np.array_equal(oracle.ca_code(prn), g[prn - 1]), prn     # LFSR tap-selector form
        assert np.array_equal(dpe.synth.ca_code(prn), g[prn - 1]), prn  # G2-delay form (product generator)


@pytest.mark.parametrize("name", ["o3_handoff_20ms", "o3_short_5ms", "o12_highrate_4ms"])
def test_o3_bcs_fft_restatement(golden, oracle, name):
    """numpy FFT restatement == pygnss vector_correlate_unfolded to fp64 rounding (O12: the same at 25 Msps)."""
    g = golden(name)
    iq, fs, S, C = g["iq"], float(g["fs"]), int(g["S"]), int(g["C"])
    assert oracle.carr_fft_len(S) == C
    raw = iq[0::2].astype(float) + 1j * iq[1::2].astype(float)
    assert np.array_equal(raw[:16], g["first16"])
    n_flip_chosen = 0
    for k in range(len(g["prn"])):
        code, carr, info = oracle.bcs_sv_fft(iq, fs, int(g["prn"][k]), g["rc"][k], g["ri"][k], g["fc"][k],
                                             g["fi"][k], int(g["cp"][k]), int(g["cp_ref"][k]))
        peak = np.abs(g["code"][k]).max()
        assert np.abs(code[S // 2 - 64:S // 2 + 65] - g["code"][k]).max() < 1e-9 * peak
        cpk = np.abs(g["carr"][k]).max()
        assert np.abs(carr[C // 2 - 256:C // 2 + 257] - g["carr"][k]).max() < 1e-9 * cpk
        # replica choice agrees with how the window was synthesised
        has_edge = 0 < info["idx_next"] < S
        if 1000 < info["idx_next"] < S - 1000 or not has_edge:   # else the choice is noise-decided
            assert info["no_flip_larger"] == (not (has_edge and bool(g["flip"][k])))
        n_flip_chosen += (not info["no_flip_larger"])
        # peak sits at lag 0 / bin 0 (channel params are the truth of the synthetic window)
        # (the unfolded correlation repeats every code period, so only the local peak is checked)
        loc = np.abs(code[S // 2 - 1000:S // 2 + 1000])
        assert loc.argmax() == 1000
        assert abs(int(np.abs(carr).argmax()) - C // 2) <= 1
    if name == "o3_handoff_20ms":
        assert n_flip_chosen == 3      # PRN 6's edge is 7 samples before the window end
    elif name == "o12_highrate_4ms":
        assert n_flip_chosen == 1      # one of the two boundaries inside the window carries a sign change
    else:
        assert n_flip_chosen == 0


@pytest.mark.parametrize("name", ["o3_handoff_20ms", "o3_short_5ms", "o12_highrate_4ms"])
def test_o3_bcs_c_oracle(golden, oracle, name):
    """C direct-sum oracle == fixtures (same maths as the FFT path, no 1/S, fftshift centre S/2)."""
    g = golden(name)
    iq, fs, C = g["iq"], float(g["fs"]), int(g["C"])
    ks = range(len(g["prn"])) if name == "o3_short_5ms" else ([0, 3] if name == "o3_handoff_20ms" else [0, 2])   # keep the CPU suite fast
    for k in ks:
        code, carr, info = oracle.bcs_sv(iq, fs, int(g["prn"][k]), g["rc"][k], g["ri"][k], g["fc"][k], g["fi"][k],
                                         int(g["cp"][k]), int(g["cp_ref"][k]), -64, 64, -40, 40, C)
        peak = np.abs(g["code"][k]).max()
        assert np.abs(code - g["code"][k]).max() < 1e-9 * peak
        cpk = np.abs(g["carr"][k]).max()
        assert np.abs(carr - g["carr"][k][256 - 40:256 + 41]).max() < 1e-9 * cpk


def test_o4_satpos(golden, oracle):
    g = golden("o4_satpos")
    for k in range(len(g["prn"])):
        st, rc = oracle.sat_pos(g["eph"][k], float(g["tx"][k]))
        assert rc == 0
        assert np.abs(st[:3] - g["sat"][k, :3]).max() < 1e-4          # metres
        assert abs(st[3] - g["sat"][k, 3]) < 1e-15                    # seconds
        assert np.abs(st[4:7] - g["sat"][k, 4:7]).max() < 1e-7        # m/s
        assert abs(st[7] - g["sat"][k, 7]) < 1e-20


def test_o5_frames(golden, oracle):
    g = golden("o5_frames")
    ll = oracle.ecef2ll(g["X_ECEF"])
    assert abs(ll[0] - float(g["lat"])) < 1e-10 and abs(ll[1] - float(g["lon"])) < 1e-14
    R = oracle.enu2ecef(ll).reshape(3, 3)
    assert np.abs(R - g["R_ECEF2ENU"].T).max() < 1e-10


def test_o6_spread_grid(golden):
    g = golden("o6_spread_grid")
    pos, vel = dpe.synth.spread_grid()
    assert np.array_equal(pos[:, :3].T, g["dX"]) and np.array_equal(pos[:, 3], g["dT"])
    assert np.array_equal(vel[:, :3].T, g["dXdot"]) and np.array_equal(vel[:, 3], g["dTdot"])
