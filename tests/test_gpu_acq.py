"""GPU: coarse acquisition (dpe_acq_*, rocFFT) vs the fp64 oracle and vs fixture O8 (the reference's
Correlator.coarse_acquisition).  fp32 FFTs of length 25000: surface within 2e-5 of the peak; identical
peak cell, hence identical rc / fi / fc; cppr / cppm within 1e-4 relative."""
import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe

pytestmark = pytest.mark.gpu
PRNS = [2, 12, 19, 28, 5, 30]


@pytest.mark.parametrize("mode,coherent", [("coherent", True), ("noncoherent", False)])
def test_acq_vs_reference_fixture_and_oracle(golden, oracle, mode, coherent):
    import torch
    g = golden("o8_acquisition")
    fs, S = float(g["fs"]), int(g["S"])
    bins = g["bins_coh"] if coherent else g["bins_non"]
    acq = dpe.Acquisition(fs, S, PRNS, bins, mode=mode, prn_chunk=4)
    acq.search(torch.from_numpy(g["iq"]).to("cuda:0"))
    res = acq.results()
    surf = acq.read_surface()
    for i in range(int(g["ncases"])):
        if bool(g["c%d_coherent" % i]) != coherent:
            continue
        prn = int(g["c%d_prn" % i])
        p = PRNS.index(prn)
        r = res[p]
        ref = oracle.coarse_acquisition(g["iq"], fs, prn, bins, coherent=coherent)
        peak = ref["surface"].max()
        assert np.abs(surf[p] - ref["surface"]).max() < 2e-5 * peak
        # fixture (pygnss) values
        assert r["max_code_idx"] == int(g["c%d_ci" % i]) and r["max_dopp_idx"] == int(g["c%d_di" % i])
        assert abs(r["rc"] - float(g["c%d_rc" % i])) < 1e-9 and r["fi"] == float(g["c%d_fi" % i])
        assert abs(r["fc"] - float(g["c%d_fc" % i])) < 1e-6
        assert abs(r["cppr"] / float(g["c%d_cppr" % i]) - 1) < 1e-4 and abs(r["cppm"] / float(g["c%d_cppm" % i]) - 1) < 1e-4
        assert r["found"] == bool(g["c%d_found" % i])
    acq.close()


def test_acq_textbook_mode_vs_oracle(golden, oracle):
    """1 ms coherent x 10 non-coherent (BASELINE.json config 5 wording): parity unpinned by the
    reference; checked against the oracle's restatement only."""
    import torch
    g = golden("o8_acquisition")
    fs, S = float(g["fs"]), int(g["S"])
    bins = oracle.acq_bins(False)          # 500 Hz raster suits a 1 ms coherent length
    acq = dpe.Acquisition(fs, S, PRNS, bins, mode="textbook")
    acq.search(torch.from_numpy(g["iq"]).to("cuda:0"))
    res = acq.results()
    surf = acq.read_surface()
    for p, prn in enumerate(PRNS):
        ref = oracle.coarse_acquisition(g["iq"], fs, prn, bins, mode="textbook")
        assert np.abs(surf[p] - ref["surface"]).max() < 2e-5 * ref["surface"].max()
        assert res[p]["max_code_idx"] == ref["max_code_idx"] and res[p]["max_dopp_idx"] == ref["max_dopp_idx"]
        assert res[p]["found"] == ref["found"]
        if prn in list(g["truth_prn"]):
            assert res[p]["found"]          # all four present SVs are found on the 500 Hz raster in this mode
    acq.close()


@pytest.mark.parametrize("mode,nbins,step", [("coherent", 125, 100.0), ("noncoherent", 125, 100.0), ("textbook", 125, 100.0),
                                             ("noncoherent", 25, 500.0)])
def test_acq_32_prns_full_search(oracle, mode, nbins, step):
    """BASELINE.json configs[4] shape, exactly as bench.py's acquisition line runs it: 32 PRNs x 125 bins (100 Hz) x 2500 delays, a
    10 ms window, all PRNs in one chunk -- in the three modes -- and the reference's non-coherent mode on the raster the reference
    pairs with it (25 x 500 Hz, correlator.py:13).  The |.| surface of the six simulated PRNs and of two absent ones against the
    oracle's restatement of coarse_acquisition (correlator.py:53-103; the textbook form is the oracle's own, parity unpinned by the
    reference), peak cell and decision for every one of the 32; the simulated PRNs found within one bin / half a chip."""
    import torch
    fs, S = 2.5e6, 25000
    ch = dpe.synth.random_channels(77, 6, prns=[3, 7, 11, 18, 22, 31])
    ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(78, fs, S, ch, amp=150.0, flip=np.zeros(6, dtype=bool))
    bins = (np.arange(nbins) - nbins // 2) * step
    acq = dpe.Acquisition(fs, S, list(range(1, 33)), bins, mode=mode, prn_chunk=32)
    acq.search(torch.from_numpy(iq).to("cuda:0"))
    res = acq.results()
    surf = acq.read_surface()
    acq.close()
    assert surf.shape == (32, nbins, 2500)
    for prn in (3, 7, 11, 18, 22, 31, 1, 32):
        kw = dict(mode="textbook") if mode == "textbook" else dict(coherent=(mode == "coherent"))
        ref = oracle.coarse_acquisition(iq, fs, prn, bins, wrap_mask=True, **kw)
        r = res[prn - 1]
        assert r["prn"] == prn
        assert np.abs(surf[prn - 1] - ref["surface"]).max() < 2e-5 * ref["surface"].max(), prn
        assert r["max_code_idx"] == ref["max_code_idx"] and r["max_dopp_idx"] == ref["max_dopp_idx"] and r["found"] == ref["found"]
        assert abs(r["cppm"] / ref["cppm"] - 1) < 1e-4 and abs(r["cppr"] / ref["cppr"] - 1) < 1e-4
    if step == 100.0:
        found = sorted(r["prn"] for r in res if r["found"])
        assert set([3, 7, 11, 18, 22, 31]) <= set(found)
        if mode == "coherent":
            assert found == [3, 7, 11, 18, 22, 31]
        for r in res:
            if r["found"] and r["prn"] in (3, 7, 11, 18, 22, 31):
                k = list(ch["prn"]).index(r["prn"])
                d = (r["rc"] - ch["rc"][k] + 511.5) % 1023 - 511.5
                assert abs(d) < 0.5 and abs(r["fi"] - ch["fi"][k]) < 100.0   # within one 100 Hz bin


def test_fused_coherent_search_equals_the_rocfft_chain():
    """The coherent search at 2 500 delays per code period runs in one fused kernel (spectrum product, a hand-written
    2 500-point inverse transform in LDS, magnitude, column maximum); DPE_ACQ_NO_FUSED=1 at create keeps the rocFFT chain
    (multiply kernel, rocFFT, fold kernel).  Same surface to fp32 rounding, same maximum row, same peak cells and statistics
    -- over 32 PRNs x 125 bins, with a bin count that leaves the last block of bins partly filled (125 = 20 x 6 + 5)."""
    import os
    import torch
    fs, S = 2.5e6, 25000
    ch = dpe.synth.random_channels(91, 5, prns=[4, 9, 16, 23, 29])
    ch["cp_ref"] = ch["cp"].copy()
    iq = torch.from_numpy(dpe.synth.gen_iq(92, fs, S, ch, amp=120.0, flip=np.zeros(5, dtype=bool))).to("cuda:0")
    bins = np.arange(-62, 63) * 100.0
    out = {}
    for form in ("fused", "rocfft"):
        old = os.environ.get("DPE_ACQ_NO_FUSED")
        if form == "rocfft":
            os.environ["DPE_ACQ_NO_FUSED"] = "1"
        try:
            acq = dpe.Acquisition(fs, S, list(range(1, 33)), bins, mode="coherent")   # the switch is read at create
        finally:
            if form == "rocfft":
                if old is None:
                    os.environ.pop("DPE_ACQ_NO_FUSED", None)
                else:
                    os.environ["DPE_ACQ_NO_FUSED"] = old
        acq.search(iq)
        out[form] = (acq.results(), acq.read_surface().copy())
        acq.close()
    (r0, s0), (r1, s1) = out["fused"], out["rocfft"]
    assert s0.shape == s1.shape == (32, 125, 2500)
    assert np.abs(s0 - s1).max() < 3e-6 * s1.max()
    for a, b in zip(r0, r1):
        assert a["max_code_idx"] == b["max_code_idx"] and a["max_dopp_idx"] == b["max_dopp_idx"] and a["found"] == b["found"]
        assert abs(a["cppm"] / b["cppm"] - 1) < 1e-5 and abs(a["cppr"] / b["cppr"] - 1) < 1e-5
    assert {4, 9, 16, 23, 29} <= {r["prn"] for r in r0 if r["found"]}   # (cppm > 2 also lets a cross-correlation peak through: both forms alike)


def test_fused_noncoherent_search_equals_the_rocfft_chain():
    """The reference's non-coherent mode (coherent = False: one 25 000-point correlation per bin, |.| summed over the ten lag
    aliases, correlator.py:77-82) at 10 x 2 500 samples runs as ten packed 2 500-point transforms per (PRN, bin) with the ten-point
    stage across them at the output, fed by a forward kernel that writes the decimated spectra (csrc/dpe_acq_pack.h);
    DPE_ACQ_NO_FUSED=1 keeps wipe kernel, multiply kernel, 25 000-point rocFFTs and fold kernel.  Same surface to fp32 rounding,
    same peak cells and statistics -- 32 PRNs x the reference's 25 x 500 Hz raster; PRNs in one launch (XCD-aware item order), in
    chunks of 8, of 5 (a short last chunk, plain item order) and of 1; with the forward transform left to rocFFT
    (DPE_ACQ_NO_FWD_PACK=1) and with the radix-10 kernel + four-pass transforms of round 4 (DPE_ACQ_NO_PACK=1)."""
    import os
    import torch
    fs, S = 2.5e6, 25000
    ch = dpe.synth.random_channels(93, 5, prns=[4, 9, 16, 23, 29])
    ch["cp_ref"] = ch["cp"].copy()
    iq = torch.from_numpy(dpe.synth.gen_iq(94, fs, S, ch, amp=120.0, flip=np.zeros(5, dtype=bool))).to("cuda:0")
    bins = np.arange(-12, 13) * 500.0
    out = {}
    forms = (("fused", 32, None), ("fused8", 8, None), ("fused5", 5, None), ("fused1", 1, None), ("rocfft_fwd", 32, "DPE_ACQ_NO_FWD_PACK"),
             ("round4", 8, "DPE_ACQ_NO_PACK"), ("rocfft", 8, "DPE_ACQ_NO_FUSED"))
    for form, chunk, env in forms:
        old = os.environ.get(env) if env else None
        if env:
            os.environ[env] = "1"
        try:
            acq = dpe.Acquisition(fs, S, list(range(1, 33)), bins, mode="noncoherent", prn_chunk=chunk)
        finally:
            if env:
                if old is None:
                    os.environ.pop(env, None)
                else:
                    os.environ[env] = old
        acq.search(iq)
        acq.search(iq)   # (a second search on the same handle: the per-delay maxima are cleared by the first kernel of a search)
        out[form] = (acq.results(), acq.read_surface().copy())
        acq.close()
    (r0, s0), (r1, s1) = out["fused"], out["rocfft"]
    assert s0.shape == s1.shape == (32, 25, 2500)
    for form in ("fused8", "fused5", "fused1"):
        assert np.array_equal(s0, out[form][1]), form
    assert np.abs(s0 - s1).max() < 3e-6 * s1.max()
    assert np.abs(out["rocfft_fwd"][1] - s1).max() < 3e-6 * s1.max()
    assert np.abs(out["round4"][1] - s1).max() < 3e-6 * s1.max()
    for form in ("fused", "rocfft_fwd", "round4"):
        for a, b in zip(out[form][0], r1):
            assert a["max_code_idx"] == b["max_code_idx"] and a["max_dopp_idx"] == b["max_dopp_idx"] and a["found"] == b["found"], form
            assert abs(a["cppm"] / b["cppm"] - 1) < 1e-5 and abs(a["cppr"] / b["cppr"] - 1) < 1e-5, form
    assert {4, 9, 23, 29} <= {r["prn"] for r in r0 if r["found"]}   # (PRN 16, the weakest, stays below cppm = 2 on this raster in both forms)


@pytest.mark.parametrize("fs", [4.0e6, 5.0e6])
@pytest.mark.parametrize("mode", ["coherent", "textbook", "noncoherent"])
def test_fused_search_at_4_and_5_msps(oracle, fs, mode):
    """Correlator.coarse_acquisition is rate-agnostic (correlator.py:53-103); at 4 000 / 5 000 samples per code period the fused
    search runs through the generic four-pass transform of csrc/dpe_acq_mixed.h (the reference's non-coherent mode: radix-10 kernel
    + ten such transforms per (PRN, bin)).  Against the oracle (surface within 2e-5 of the
    peak, same cells, statistics within 1e-4) for present and absent PRNs, and against the rocFFT chain (DPE_ACQ_NO_FUSED=1) on
    32 PRNs x 31 bins (a short last block of bins): surface within 3e-6 of the peak, identical cells."""
    import os
    import torch
    n_ms = 10
    S = int(round(fs * 1e-3)) * n_ms
    present = [6, 13, 21, 27]
    ch = dpe.synth.random_channels(311, 4, prns=present)
    step = 100.0 if mode == "coherent" else 500.0
    bins = (np.arange(31) - 15) * step
    ch["fi"] = np.array([-3.2, 1.7, 9.4, -11.1]) * step
    ch["fc"] = 1.023e6 * (1.0 + ch["fi"] / 1.57542e9)
    ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(312, fs, S, ch, amp=150.0, flip=np.zeros(4, dtype=bool))
    d = torch.from_numpy(iq).to("cuda:0")
    prns = list(range(1, 33))
    out = {}
    for form in ("fused", "rocfft"):
        old = os.environ.get("DPE_ACQ_NO_FUSED")
        if form == "rocfft":
            os.environ["DPE_ACQ_NO_FUSED"] = "1"
        try:
            acq = dpe.Acquisition(fs, S, prns, bins, mode=mode, prn_chunk=8)
        finally:
            if form == "rocfft":
                if old is None:
                    os.environ.pop("DPE_ACQ_NO_FUSED", None)
                else:
                    os.environ["DPE_ACQ_NO_FUSED"] = old
        acq.search(d)
        acq.search(d)
        out[form] = (acq.results(), acq.read_surface().copy())
        acq.close()
    (r0, s0), (r1, s1) = out["fused"], out["rocfft"]
    assert s0.shape == s1.shape == (32, 31, S // n_ms)
    assert np.abs(s0 - s1).max() < 3e-6 * s1.max()
    for a, b in zip(r0, r1):
        assert a["max_code_idx"] == b["max_code_idx"] and a["max_dopp_idx"] == b["max_dopp_idx"] and a["found"] == b["found"]
        assert abs(a["cppm"] / b["cppm"] - 1) < 1e-5 and abs(a["cppr"] / b["cppr"] - 1) < 1e-5
    if mode != "noncoherent":   # (the magnitude sum over ten aliases raises the floor: a PRN between two 500 Hz bins stays below cppm = 2 in both forms and in the oracle)
        assert set(present) <= {r["prn"] for r in r0 if r["found"]}
    for prn in present + [2, 30]:
        q = prns.index(prn)
        ref = oracle.coarse_acquisition(iq, fs, prn, bins, coherent=(mode == "coherent"), mode="textbook" if mode == "textbook" else None)
        assert np.abs(s0[q] - ref["surface"]).max() < 2e-5 * ref["surface"].max()
        assert (r0[q]["max_code_idx"], r0[q]["max_dopp_idx"]) == (ref["max_code_idx"], ref["max_dopp_idx"])
        assert abs(r0[q]["cppm"] / ref["cppm"] - 1) < 1e-4 and abs(r0[q]["cppr"] / ref["cppr"] - 1) < 1e-4
        assert r0[q]["found"] == ref["found"]


def test_acq_statistics_with_a_long_delay_row(oracle):
    """16.368 Msps x 1 ms: 16 368 delays per PRN -- the statistics kernel's row no longer fits the LDS a launch gets without an
    opt-in beside its 17 KB of static LDS and is read from memory instead.  Peak cell and statistics against the
    oracle."""
    import torch
    fs, S = 16.368e6, 16368
    ch = dpe.synth.random_channels(95, 2, prns=[6, 21])
    ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(96, fs, S, ch, amp=400.0, flip=np.zeros(2, dtype=bool))
    bins = np.arange(-8, 9) * 500.0
    acq = dpe.Acquisition(fs, S, [6, 21, 13], bins, mode="coherent")
    acq.search(torch.from_numpy(iq).to("cuda:0"))
    res = acq.results()
    for p, prn in enumerate([6, 21, 13]):
        ref = oracle.coarse_acquisition(iq, fs, prn, bins, coherent=True)
        assert res[p]["max_code_idx"] == ref["max_code_idx"] and res[p]["max_dopp_idx"] == ref["max_dopp_idx"]
        assert res[p]["found"] == ref["found"]      # (cppm > 2 also passes the absent PRN 13 on a 1 ms window: reference and HIP path alike)
        assert abs(res[p]["cppm"] / ref["cppm"] - 1) < 2e-4 and abs(res[p]["cppr"] / ref["cppr"] - 1) < 2e-4
    acq.close()


def test_acq_statistics_of_a_nearly_constant_row(oracle):
    """The statistics kernel selects its two order statistics through one histogram of linear bins (mean / 256 wide) and ranks
    the few values of the selected bins inside a wave; rows whose selected bin holds more than 64 values fall back to the radix
    passes.  A DC window (+-1 LSB of noise) is such a row: the 0 Hz bin's correlation is the same for every delay (DC x the code's chip sum),
    so the 2500 values of max_percode differ by a few percent and share some ten bins.  cppr and cppm (both ~ 1) against the oracle; nothing is found."""
    import torch
    fs, S = 2.5e6, 25000
    iq = np.random.default_rng(5).integers(-1, 2, size=2 * S).astype(np.int16)   # +-1 LSB of noise: the values differ, in the 4th digit
    iq[0::2] += 1000
    iq[1::2] -= 300
    bins = np.arange(-4, 5) * 100.0
    acq = dpe.Acquisition(fs, S, [1, 9, 30], bins, mode="coherent")
    acq.search(torch.from_numpy(iq).to("cuda:0"))
    res = acq.results()
    for p, prn in enumerate([1, 9, 30]):
        ref = oracle.coarse_acquisition(iq, fs, prn, bins, coherent=True)
        assert not res[p]["found"] and not ref["found"]
        assert abs(res[p]["cppm"] / ref["cppm"] - 1) < 2e-4 and abs(res[p]["cppr"] / ref["cppr"] - 1) < 2e-4
        assert 0.9 < res[p]["cppm"] < 1.1      # (a few percent of spread: some 250 values per bin)
    acq.close()


@pytest.mark.gpu
def test_o9_fine_frequency_and_two_window_driver(golden):
    """HIP search_signal (coarse + fine frequency) and the two-window driver against the reference's own
    outputs (fixture O9) and against the oracle.  The fine stage runs a 262144-point fp32 FFT: the peak bin
    must be the reference's, ri within 2e-5 cycles, fi / fc exact functions of the bin."""
    import torch
    from oracle import oracle as o
    g = golden("o9_scalar_acquisition")
    fs, S = float(g["fs"]), int(g["S"])
    prns = [int(p) for p in g["prn_list"]]
    iq = np.ascontiguousarray(g["iq"])
    iq_d = torch.from_numpy(iq).to("cuda:0")
    w0, w1 = iq_d[:2 * S], iq_d[2 * S:]
    acq = dpe.Acquisition(fs, S, prns, g["bins"], mode="coherent")
    ref = g["per_window"]                               # [2][P][found, rc, ri, fc, fi, cppr, cppm]
    for w, buf in enumerate((w0, w1)):
        got = acq.search_signal(buf)
        for i, r in enumerate(got):
            assert r["found"] == bool(ref[w, i, 0])
            assert abs(r["rc"] - ref[w, i, 1]) < 1e-9
            assert abs(r["fi"] - ref[w, i, 4]) < 1e-9 and abs(r["fc"] - ref[w, i, 3]) < 1e-6      # same FFT bin
            d = abs(r["ri"] - ref[w, i, 2])
            assert min(d, 1.0 - d) < 2e-5
            assert abs(r["cppr"] / ref[w, i, 5] - 1) < 2e-4 and abs(r["cppm"] / ref[w, i, 6] - 1) < 2e-4
            orc = o.search_signal(iq[2 * S * w:2 * S * (w + 1)], fs, prns[i])
            assert orc["max_carr_idx"] == r["max_carr_idx"]
    fin = acq.scalar_acquisition(w0, w1)
    for i, r in enumerate(fin):
        assert r["from_second_window"] == bool(ref[1, i, 6] > ref[0, i, 6])
        assert abs(r["rc"] - g["final"][i, 0]) < 1e-6 and abs(r["fc"] - g["final"][i, 2]) < 1e-6
        assert abs(r["fi"] - g["final"][i, 3]) < 1e-9
        d = abs(r["ri"] - g["final"][i, 1])
        assert min(d, 1.0 - d) < 2e-5
    acq.close()


@pytest.mark.parametrize("fs", [4.0e6, 5.0e6])
def test_search_signal_at_4_and_5_msps(oracle, fs):
    """Correlator.search_signal (correlator.py:38-51: coarse, then fine frequency) away from the reference's rate: the fused coarse
    search of csrc/dpe_acq_mixed.h feeding the fine-frequency stage (zero-padded 8 << S.bit_length() point FFT).  Against the
    oracle: same coarse cell, same fine FFT bin (hence rc / fi / fc exact), ri within 2e-5 cycles, statistics within 2e-4."""
    import torch
    S = int(round(fs * 1e-2))
    present = [5, 14, 23]
    ch = dpe.synth.random_channels(411, 3, prns=present)
    bins = (np.arange(41) - 20) * 100.0
    ch["fi"] = np.array([-1234.0, 310.0, 1777.0])
    ch["fc"] = 1.023e6 * (1.0 + ch["fi"] / 1.57542e9)
    ch["cp_ref"] = ch["cp"].copy()
    iq = dpe.synth.gen_iq(412, fs, S, ch, amp=150.0, flip=np.zeros(3, dtype=bool))
    prns = present + [9]
    acq = dpe.Acquisition(fs, S, prns, bins, mode="coherent")
    got = acq.search_signal(torch.from_numpy(iq).to("cuda:0"))
    acq.close()
    for r, prn in zip(got, prns):
        ref = oracle.search_signal(iq, fs, prn, bins=bins, coherent=True)
        assert r["found"] == ref["found"] and (prn not in present or r["found"])
        assert (r["max_code_idx"], r["max_dopp_idx"]) == (ref["max_code_idx"], ref["max_dopp_idx"])
        assert r["max_carr_idx"] == ref["max_carr_idx"]
        assert abs(r["rc"] - ref["rc"]) < 1e-9 and abs(r["fi"] - ref["fi"]) < 1e-9 and abs(r["fc"] - ref["fc"]) < 1e-6
        d = abs(r["ri"] - ref["ri"])
        assert min(d, 1.0 - d) < 2e-5
        assert abs(r["cppr"] / ref["cppr"] - 1) < 2e-4 and abs(r["cppm"] / ref["cppm"] - 1) < 2e-4

