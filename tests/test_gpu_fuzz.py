"""Randomised differential test of the HIP path against the CPU oracle: window lengths that are not multiples of any
tile size, 1..37 SVs, 1..5 windows, ragged grids of different sizes, every lag/bin half-width family of the kernels.
Seeded: the default 96 + 36 + 18 cases always are the same ones (round 6: four times the earlier defaults, ~25 s more); DPE_FUZZ_CASES / DPE_FUZZ_SEED widen or move the sweep
(`DPE_FUZZ_CASES=600 python -m pytest tests/test_gpu_fuzz.py -m gpu -n 4` is the long form; ~10 000 cases were run in round 1).

Tolerance 2e-5 of the peak / maximum score (DPE_FUZZ_TOL), against 2e-6 in the named parity tests: the sweep mixes in what
those avoid on purpose -- a single weak SV (the "peak" is then close to the fp32 rounding of the noise it is summed out of),
lpower = 2 (relative errors double) and lag windows beyond +-32 (boundary-difference kernel).  Worst case seen in ~10 000 cases: 1.14e-5 (one weak
SV, 4330-sample window, lpower 2, L = 70); every case that exceeded 2e-6 had L > 32 or lpower = 2.  Arg-max, nav-bit decisions, out-of-window counts and the DC mean
stay exact."""
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("DPE_FUZZ_CASES", "96"))
SEED = int(os.environ.get("DPE_FUZZ_SEED", "1234"))


def draw(i):
    rng = np.random.Generator(np.random.PCG64(SEED * 100003 + i))
    fs = float(rng.choice([2.046e6, 2.5e6, 4.0e6, 5.0e6]))
    S = 2 * int(rng.integers(1024, 15001))      # even: the reference's fftshift centre S/2 (dpe_bcm_create refuses odd S)
    if rng.random() < 0.25:
        S = int(rng.choice([4096, 8192, 16384, 12500, 25000, 20460, 50000, 40920]))
    K = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 13, 20, 37, int(rng.integers(1, 38))]))
    W = int(rng.choice([1, 1, 2, 3, 5, 8]))
    G = int(rng.choice([1, 7, 255, 256, 1023, 1024, 1025, int(rng.integers(2, 6000))]))
    vel_G = int(rng.choice([G, 1, 1024, int(rng.integers(2, 6000))]))
    # metres of grid half-width / c * fs = lags the position grid can reach; keep the bank wide enough
    need_L = int(np.ceil(140.0 / 299792458.0 * fs)) + 2
    L = int(rng.choice([need_L, need_L + 1, 8, 16, 17, 32, 33, 40, 70]))
    L = max(L, need_L)
    # widest B the moment expansion takes at this C (include/dpe_hip.h, dpe_bcs_config.binHalfWidth)
    C = 8 * (1 << int(np.ceil(np.log2(S))))
    b_max = int(np.floor((720 * 2e-7) ** (1.0 / 6.0) * C / (2 * np.pi * 127.5) * 0.999))
    B = min(int(rng.choice([12, 20, 32, 33, 64, b_max])), b_max)
    B = min(B, (150 * 1024 // K - 32) // 16 // 2 - 1)   # all K banks must fit the scan's LDS (dpe_bcm_config.binHalfWidth)
    lpower = int(rng.choice([1, 1, 2]))
    amp = float(rng.choice([48.0, 200.0]))
    # drawn after everything else so that the first round's sweep keeps its parameters
    grid, offset = "rand", None
    u = rng.random()
    if u < 0.15:                       # banks narrower than the grids reach: the clamped (zero-slot) variants of the scan
        L, B = int(rng.integers(1, 3)), min(B, int(rng.integers(1, 3)))
    elif u < 0.30:                     # Cartesian grid, x slowest / t fastest (BCM_InitPosGrid order)
        grid, G = "uniform", int(rng.integers(2, 9)) ** 4
        vel_G = None
    if rng.random() < 0.3:             # grid centre away from the truth (ENU metres, clock metres)
        offset = [float(v) for v in rng.uniform(-40.0, 40.0, 4)]
        L = L if u < 0.15 else max(L, need_L + int(np.ceil(70.0 / 299792458.0 * fs)))
    # handle options: arg-max only (no score write), with / without the weighted-mean sums, Update issued twice
    ws, wm, rep = bool(rng.random() < 0.8), bool(rng.random() < 0.5), int(rng.choice([1, 1, 2]))
    return dict(seed=1000 + i, fs=fs, S=S, K=K, W=W, G=G, vel_G=vel_G, L=L, B=B, lpower=lpower, amp=amp, grid=grid,
                offset=offset, write_scores=ws, weighted_mean=wm, repeats=rep)


@pytest.mark.parametrize("i", range(N_CASES))
def test_random_case(i):
    p = draw(i)
    case = helpers.make_case(seed=p["seed"], fs=p["fs"], S=p["S"], K=p["K"], G=p["G"], vel_G=p["vel_G"], amp=p["amp"],
                             W=p["W"], grid=p["grid"], center_offset=p["offset"])
    try:
        out = helpers.run_gpu(case, p["L"], p["B"], lpower=p["lpower"], write_scores=p["write_scores"],
                              weighted_mean=p["weighted_mean"], repeats=p["repeats"])
        ref = helpers.run_oracle(case, p["L"], p["B"], lpower=p["lpower"])
        # one or two SVs on a handful of points: nothing averages the reference's own index noise (1.4e-4 samples per pair)
        helpers.assert_parity(out, ref, tol=float(os.environ.get('DPE_FUZZ_TOL', '2e-5')), pos_ref_noise=3e-4,
                              check_scores=p["write_scores"])
    except Exception:
        print("fuzz case %d: %r" % (i, p))
        raise


N_WIDE = int(os.environ.get("DPE_FUZZ_WIDE_CASES", "36"))


def draw_wide(i):
    """Round 2: the high-rate / wide-window forms of stage 1 -- chip-boundary kernel (4 ... 30 Msps, lag windows of 17 ... 31
    samples and chunked ones beyond), boundary-difference and dense fall-backs (carrier offsets beyond the chip kernel's
    closed-form range), and the full-length FFT form (lag windows beyond +-292, bin windows beyond the moment expansion)."""
    rng = np.random.Generator(np.random.PCG64((SEED + 7) * 100003 + i))
    fs = float(rng.choice([4.0e6, 5.0e6, 8.0e6, 10.0e6, 12.5e6, 16.0e6, 20.0e6, 25.0e6, 30.0e6]))
    S = 2 * int(rng.integers(2200, 60001))
    K = int(rng.choice([1, 2, 3, 5, 8, 12]))
    W = int(rng.choice([1, 1, 2, 3]))
    G = int(rng.choice([1, 255, 1024, int(rng.integers(2, 3000))]))
    need_L = int(np.ceil(140.0 / 299792458.0 * fs)) + 2
    L = max(need_L, int(rng.choice([17, 20, 24, 28, 31, 31, 32, 40, 64, 100])))
    C = 8 * (1 << int(np.ceil(np.log2(S))))
    b_max = int(np.floor((720 * 2e-7) ** (1.0 / 6.0) * C / (2 * np.pi * 127.5) * 0.999))
    B = min(int(rng.choice([8, 12, 20, 33])), b_max)
    u = rng.random()
    if u < 0.12:
        L = int(rng.choice([300, 400]))            # FFT form (lag window)
    elif u < 0.24:
        B = min(b_max + int(rng.integers(1, 40)), 200)   # FFT form (bin window)
    L = min(L, S // 2 - 300)
    offset = [float(v) for v in rng.uniform(-40.0, 40.0, 4)] if rng.random() < 0.3 else None
    if offset is not None:
        L = max(L, need_L + int(np.ceil(70.0 / 299792458.0 * fs)))
    return dict(seed=5000 + i, fs=fs, S=S, K=K, W=W, G=G, L=L, B=B, amp=float(rng.choice([48.0, 200.0])), offset=offset,
                lpower=int(rng.choice([1, 1, 2])), if_offset=float(rng.choice([0.0, 0.0, 0.0, 60e3])))


@pytest.mark.parametrize("i", range(N_WIDE))
def test_random_high_rate_or_wide_window_case(i):
    import navlab_dpe_sdr_amd as dpe
    p = draw_wide(i)
    case = helpers.make_case(seed=p["seed"], fs=p["fs"], S=p["S"], K=p["K"], G=p["G"], amp=p["amp"], W=p["W"],
                             center_offset=p["offset"])
    if p["if_offset"]:                 # an intermediate frequency: outside the chip kernel's closed-form DC term -> fall-back kernels
        for w in case["wins"]:
            w["start"]["fi"] = w["start"]["fi"] + p["if_offset"]
            w["fi"] = w["fi"] + p["if_offset"]
            w["iq"] = dpe.synth.gen_iq(p["seed"] * 1000 + 1, case["fs"], case["S"], dict(w["start"]), amp=p["amp"], flip=w["flip"])
    try:
        out = helpers.run_gpu(case, p["L"], p["B"], lpower=p["lpower"], weighted_mean=False)
        ref = helpers.run_oracle(case, p["L"], p["B"], lpower=p["lpower"])
        helpers.assert_parity(out, ref, tol=float(os.environ.get('DPE_FUZZ_TOL', '2e-5')), pos_ref_noise=3e-4)
    except Exception:
        print("wide fuzz case %d: %r" % (i, p))
        raise


N_ACQ = int(os.environ.get("DPE_FUZZ_ACQ_CASES", "18"))


def draw_acq(i):
    rng = np.random.Generator(np.random.PCG64(SEED * 7919 + i))
    fs = float(rng.choice([2.046e6, 2.5e6, 4.0e6, 5.0e6]))
    n_ms = int(rng.choice([2, 4, 5, 10]))
    mode = str(rng.choice(["coherent", "noncoherent", "textbook"]))
    present = sorted(int(p) for p in rng.choice(np.arange(1, 33), size=int(rng.integers(1, 5)), replace=False))
    absent = sorted(int(p) for p in rng.choice([p for p in range(1, 38) if p not in present],
                                               size=int(rng.integers(0, 4)), replace=False))
    if mode == "coherent":          # a coherent sum over n_ms ms resolves 1/(n_ms ms): raster of half that
        step = float(rng.choice([50.0, 100.0])) * 10.0 / n_ms
    else:
        step = float(rng.choice([250.0, 500.0]))
    nb = int(rng.integers(3, 40)) | 1
    return dict(seed=5000 + i, fs=fs, S=int(round(fs * 1e-3)) * n_ms, mode=mode, present=present, absent=absent, step=step,
                nb=nb, chunk=int(rng.choice([0, 1, 3, 32])), amp=float(rng.choice([100.0, 200.0])))


@pytest.mark.parametrize("i", range(N_ACQ))
def test_random_acquisition_case(i):
    """Coarse acquisition vs the oracle's restatement of Correlator.coarse_acquisition: window lengths of 2..10 code
    periods at four sampling rates, all three modes, random PRN sets (present and absent), rasters and chunk sizes."""
    import torch
    import navlab_dpe_sdr_amd as dpe
    o = helpers._oracle()
    p = draw_acq(i)
    try:
        fs, S = p["fs"], p["S"]
        K = len(p["present"])
        ch = dpe.synth.random_channels(p["seed"], K, prns=p["present"])
        bins = (np.arange(p["nb"]) - p["nb"] // 2) * p["step"]
        ch["fi"] = np.random.Generator(np.random.PCG64(p["seed"] + 1)).uniform(bins[0], bins[-1], K)   # inside the raster
        ch["fc"] = 1.023e6 * (1.0 + ch["fi"] / 1.57542e9)
        ch["cp_ref"] = ch["cp"].copy()
        iq = dpe.synth.gen_iq(p["seed"] + 2, fs, S, ch, amp=p["amp"], flip=np.zeros(K, dtype=bool))
        prns = p["present"] + p["absent"]
        acq = dpe.Acquisition(fs, S, prns, bins, mode=p["mode"], prn_chunk=p["chunk"])
        acq.search(torch.from_numpy(iq).to("cuda:0"))
        res, surf = acq.results(), acq.read_surface()
        acq.close()
        for q, prn in enumerate(prns):
            kw = dict(coherent=(p["mode"] == "coherent"), mode="textbook" if p["mode"] == "textbook" else None)
            try:
                ref = o.coarse_acquisition(iq, fs, prn, bins, **kw)
            except IndexError:
                # peak within ceil(fs/F_CA) delays of the last one: the reference's mask indexes past the end and raises
                # (correlator.py:96-99); the HIP path wraps the mask at both ends
                ref = o.coarse_acquisition(iq, fs, prn, bins, wrap_mask=True, **kw)
            peak = ref["surface"].max()
            assert np.abs(surf[q] - ref["surface"]).max() < 2e-5 * peak, "surface PRN %d" % prn
            r = res[q]
            if (r["max_code_idx"], r["max_dopp_idx"]) != (ref["max_code_idx"], ref["max_dopp_idx"]):   # fp32 tie only
                assert ref["surface"][r["max_dopp_idx"], r["max_code_idx"]] > peak * (1 - 2e-5)
            else:
                assert abs(r["rc"] - ref["rc"]) < 1e-9 and r["fi"] == ref["fi"]
                assert abs(r["cppr"] / ref["cppr"] - 1) < 1e-4 and abs(r["cppm"] / ref["cppm"] - 1) < 1e-4
                if abs(ref["cppm"] - 2.0) > 1e-3:
                    assert r["found"] == ref["found"]
    except Exception:
        print("acquisition fuzz case %d: %r" % (i, p))
        raise
