"""Full-length FFT form of stage 1 (csrc/dpe_bcs_fft.h): the fallback for code-lag windows beyond DPE_MAX_LAG_HALF_WIDTH and
Doppler windows beyond the moment expansion -- the reference's own formulation (batchcorrscores.cu:1099-1180) on hipFFT.
Against the fp64 oracle and against the streaming kernels; tolerance 2e-6 of the bank's peak."""
import os

import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe
from tests import helpers

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _banks(case, L, B, force_fft=False, max_channels=None, timing=None):
    import torch
    iq, cs, _, _ = helpers.pack_gpu_inputs(case)
    W, K = cs.shape
    if force_fft:
        os.environ["DPE_BCS_FORCE_FFT"] = "1"
    try:
        bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=case["S"], lag_half_width=L, bin_half_width=B,
                                  max_windows=W, max_channels=max_channels or K)
        bcs.Start()
    finally:
        os.environ.pop("DPE_BCS_FORCE_FFT", None)
    d = torch.from_numpy(iq).to("cuda:0")
    bcs.Update(d, cs)
    if timing is not None:          # steady-state time of one Update (plans exist after the first)
        torch.cuda.synchronize()
        bcs.profile(True)
        for _ in range(3):
            bcs.Update(d, cs)
        p = bcs.profile(False)
        timing["ms"] = sum(v[0] for v in p.values()) / 3.0
    code, carr = bcs.read_banks()
    info = bcs.read_info()
    kern = bcs.stage1_kernel
    bcs.Stop()
    return code, carr, info, kern


def _check_vs_oracle(case, L, B, code, carr, info):
    from oracle import oracle as o
    for wi, w in enumerate(case["wins"]):
        s = w["start"]
        for k in range(case["K"]):
            c, f, inf = o.bcs_sv(w["iq"], case["fs"], int(s["prn"][k]), s["rc"][k], s["ri"][k], s["fc"][k], s["fi"][k],
                                 int(s["cp"][k]), int(s["cp_ref"][k]), -L, L, -B, B, case["C"])
            assert np.abs(code[wi][k] - c).max() < TOL * np.abs(c).max()
            assert np.abs(carr[wi][k] - f).max() < TOL * np.abs(f).max()
            assert info[0][wi, k] == inf["idx_next"] and bool(info[1][wi, k]) == inf["no_flip_larger"]
        assert info[2][wi] == inf["mean"]


def test_lag_window_of_400_samples():
    """L = 400 > DPE_MAX_LAG_HALF_WIDTH: FFT path selected by itself; banks vs the oracle's direct sums."""
    case = helpers.make_case(seed=71, S=50000, K=4, G=64, amp=100.0, W=2)
    code, carr, info, kern = _banks(case, 400, 48)
    assert kern.startswith("rocfft")
    _check_vs_oracle(case, 400, 48, code, carr, info)


def test_fft_path_at_config_h_scale():
    """The fallback at BASELINE configs[2]'s lengths: S = 500000 (2^5 5^6), C = 4194304, L = 400 -- two of twelve possible
    channels tracked, so the batched plans are made for two rows per plane, not twelve.  Banks vs the oracle's direct sums."""
    case = helpers.make_case(seed=77, fs=25e6, S=500000, K=2, G=64, amp=60.0)
    assert case["C"] == 4194304
    t = {}
    code, carr, info, kern = _banks(case, 400, 16, max_channels=12, timing=t)
    assert kern.startswith("rocfft")
    print("FFT form, S = 500000, C = 4194304, 2 channels: %.3f ms per window" % t["ms"])
    _check_vs_oracle(case, 400, 16, code, carr, info)


def test_bin_window_beyond_the_moment_expansion():
    """S = 12500 -> C = 131072: the moment expansion takes B <= 37; B = 90 goes through the C-point transform."""
    case = helpers.make_case(seed=72, S=12500, K=3, G=64, amp=100.0)
    code, carr, info, kern = _banks(case, 8, 90)
    assert kern.startswith("rocfft")
    _check_vs_oracle(case, 8, 90, code, carr, info)


@pytest.mark.parametrize("kw,L,B", [
    (dict(seed=73, S=50000, K=8, G=64, amp=48.0, W=3), 4, 20),                       # the R shape, 45 dB-Hz, flips in some windows
    (dict(seed=74, fs=2.046e6, S=40920, K=5, G=64, amp=100.0), 8, 30),               # ns-rounded sample times (TABLE variants)
    (dict(seed=75, S=12502, K=2, G=64, amp=100.0, W=5), 16, 24),                     # S = 2 x 7 x 19 x 47: Bluestein lengths
])
def test_fft_path_equals_the_streaming_kernels(kw, L, B):
    case = helpers.make_case(**kw)
    code, carr, info, kern = _banks(case, L, B, force_fft=True)
    code0, carr0, info0, kern0 = _banks(case, L, B)
    assert kern.startswith("rocfft") and not kern0.startswith("rocfft")
    _check_vs_oracle(case, L, B, code, carr, info)
    for wi in range(case["W"]):
        for k in range(case["K"]):
            assert np.abs(code[wi][k] - code0[wi][k]).max() < TOL * np.abs(code0[wi][k]).max()
            assert np.abs(carr[wi][k] - carr0[wi][k]).max() < TOL * np.abs(carr0[wi][k]).max()
        assert np.array_equal(info[1][wi], info0[1][wi]) and info[2][wi] == info0[2][wi]


def test_whole_path_with_a_45_km_clock_bias_grid():
    """A position grid whose clock-bias axis spans +-45 km (a cold receiver clock): +-392 code lags at 2.5 Msps.  Banks
    through the FFT path, scan over the 801-entry banks, against the oracle."""
    case = helpers.make_case(seed=76, S=50000, K=4, G=3000, amp=200.0)
    case["pos"] = dpe.synth.rand_grid(176, 3000, half=(1000.0, 1000.0, 1000.0, 45000.0))
    case["pos"][0] = 0.0
    L, B = dpe.pipeline.bank_half_widths(case["pos"], case["vel"], case["fs"], case["C"])
    assert 292 < L <= 400
    out = helpers.run_gpu(case, L, B, weighted_mean=False)   # (the fp32 weighted-mean sums are not meant for 45 km offsets)
    ref = helpers.run_oracle(case, L, B)
    helpers.assert_parity(out, ref, tol=2e-5)
    assert ref["res"][0]["posOutOfWindow"] == 0
