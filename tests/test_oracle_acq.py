"""CPU: the acquisition oracle (oracle.coarse_acquisition) against fixture O8 = the reference's
Correlator.coarse_acquisition (pygnss/pythonreceiver/scalar/correlator.py:53-103) on a synthetic
10 ms window with four present and two absent PRNs, coherent and alias-non-coherent modes."""
import numpy as np


def _case(g, i):
    keys = ("prn", "coherent", "found", "rc", "fc", "fi", "cppr", "cppm", "ci", "di", "nb", "shape", "row_max", "col_every25")
    return {k: g["c%d_%s" % (i, k)] for k in keys}


def test_o8_coarse_acquisition(golden, oracle):
    g = golden("o8_acquisition")
    fs = float(g["fs"])
    n_found = 0
    for i in range(int(g["ncases"])):
        c = _case(g, i)
        coh = bool(c["coherent"])
        bins = g["bins_coh"] if coh else g["bins_non"]
        assert np.array_equal(bins, oracle.acq_bins(coh))
        r = oracle.coarse_acquisition(g["iq"], fs, int(c["prn"]), bins, coherent=coh)
        assert r["surface"].shape == tuple(c["shape"])
        assert r["max_code_idx"] == int(c["ci"]) and r["max_dopp_idx"] == int(c["di"])
        a = r["surface"]
        nb = a[max(r["max_dopp_idx"] - 8, 0):r["max_dopp_idx"] + 9, :][:, np.arange(r["max_code_idx"] - 8, r["max_code_idx"] + 9) % a.shape[1]]
        peak = c["nb"].max()
        assert np.abs(nb - c["nb"]).max() < 1e-9 * peak
        assert np.abs(a.max(1) - c["row_max"]).max() < 1e-9 * peak
        assert np.abs(r["max_percode"][::25] - c["col_every25"]).max() < 1e-9 * peak
        assert abs(r["rc"] - float(c["rc"])) < 1e-9 and r["fi"] == float(c["fi"]) and abs(r["fc"] - float(c["fc"])) < 1e-6
        assert abs(r["cppr"] - float(c["cppr"])) < 1e-9 * float(c["cppr"])
        assert abs(r["cppm"] - float(c["cppm"])) < 1e-9 * float(c["cppm"])
        assert r["found"] == bool(c["found"])
        n_found += r["found"]
        # present PRNs, 100 Hz raster: the estimate sits on the truth (code phase within a sample, Doppler
        # within half a bin).  (The 25 x 500 Hz raster with a 10 ms coherent correlation has nulls every
        # 100 Hz, so the reference itself misses or mislocates most signals there -- recorded, not judged.)
        if not coh:
            continue
        if int(c["prn"]) in list(g["truth_prn"]):
            k = list(g["truth_prn"]).index(int(c["prn"]))
            assert r["found"]
            d = (r["rc"] - g["truth_rc"][k] + 511.5) % 1023 - 511.5
            assert abs(d) < 1.023e6 / fs + 1e-9
            assert abs(r["fi"] - g["truth_fi"][k]) <= (bins[1] - bins[0]) / 2 + 1e-9
        else:
            assert not r["found"]
    assert n_found == 6
