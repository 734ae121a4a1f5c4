"""Chip-boundary stage-1 kernel (csrc/dpe_bcs_chip.h; high sampling rates, lag windows wider than +-16 samples) against
the fp64 oracle's banks (batchcorrscores.cu:1043-1180 semantics) and against the per-sample kernels it replaces.
Tolerance: 2e-6 of the bank's peak magnitude, nav-bit decisions and DC mean exact."""
import os

import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe
from tests import helpers

pytestmark = pytest.mark.gpu
TOL = 2e-6


def _banks(case, L, B, env=None):
    import torch
    iq, cs, _, _ = helpers.pack_gpu_inputs(case)
    W, K = cs.shape
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=case["S"], lag_half_width=L, bin_half_width=B,
                                  max_windows=W, max_channels=K)
        bcs.Start()                      # the switches are read at create
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    d = torch.from_numpy(iq).to("cuda:0")
    bcs.Update(d, cs)
    code, carr = bcs.read_banks()
    info = bcs.read_info()
    bcs.profile(True)
    bcs.Update(d, cs)
    prof = bcs.profile(False)
    prof["kernel"] = bcs.stage1_kernel
    bcs.Stop()
    return code, carr, info, prof


def _check(case, L, B, tol=TOL):
    from oracle import oracle as o
    code, carr, info, _ = _banks(case, L, B)
    code0, carr0, info0, _ = _banks(case, L, B, {"DPE_BCS_NO_CHIP": "1"})
    code1, carr1, info1, p1 = _banks(case, L, B, {"DPE_BCS_NO_CHIP2": "1"})     # the first form of the chip kernel
    assert p1["kernel"] != "bcs_bank_chip2_kernel"
    _check.last_kernels = (p1["kernel"],)
    worst = 0.0
    for wi, w in enumerate(case["wins"]):
        s = w["start"]
        for k in range(case["K"]):
            c, f, inf = o.bcs_sv(w["iq"], case["fs"], int(s["prn"][k]), s["rc"][k], s["ri"][k], s["fc"][k], s["fi"][k],
                                 int(s["cp"][k]), int(s["cp_ref"][k]), -L, L, -B, B, case["C"])
            for name, got, ref in (("code", code[wi][k], c), ("carr", carr[wi][k], f), ("code/per-sample", code0[wi][k], c), ("carr/per-sample", carr0[wi][k], f),
                                   ("code/chip", code1[wi][k], c), ("carr/chip", carr1[wi][k], f)):
                err = np.abs(got - ref).max() / np.abs(ref).max()
                worst = max(worst, err)
                assert err < tol, "%s window %d SV %d: rel err %.3g" % (name, wi, k, err)
            assert info[0][wi, k] == inf["idx_next"] and bool(info[1][wi, k]) == inf["no_flip_larger"]
            assert bool(info0[1][wi, k]) == inf["no_flip_larger"] and bool(info1[1][wi, k]) == inf["no_flip_larger"]
        assert info[2][wi] == info0[2][wi] == info1[2][wi] == inf["mean"]          # DC mean: exact integer sums in every path
    return worst


@pytest.mark.parametrize("kw,L,B", [
    (dict(seed=11, fs=25e6, S=125000, K=4, G=64, amp=60.0), 31, 16),            # 5 ms at 25 Msps: 115 passes of 1088 samples
    (dict(seed=12, fs=25e6, S=100002, K=5, G=64, amp=30.0, W=2), 24, 12),       # ragged window, two windows, 45 dB-Hz
    (dict(seed=13, fs=10e6, S=200000, K=3, G=64, amp=100.0), 20, 30),           # 9.8 samples per chip: two chip rounds per pass
    (dict(seed=14, fs=5e6, S=100000, K=9, G=64, amp=100.0), 17, 40),            # 4.9 samples per chip: four rounds
    (dict(seed=17, fs=20e6, S=400000, K=2, G=64, amp=50.0), 31, 20),            # full 20 ms window at 20 Msps
])
def test_chip_kernel_vs_oracle_and_per_sample_kernels(kw, L, B):
    worst = _check(helpers.make_case(**kw), L, B)
    print("worst rel err", worst)


def test_chip_kernel_really_runs_and_is_selected_for_config_h():
    """The H configuration must go through the chip kernel (DPE_BCS_NO_CHIP switches it off), and faster than the
    boundary-difference kernel it replaces."""
    cfg = dpe.workload.CONFIG_H
    case = helpers.make_case(seed=21, fs=cfg["fs"], S=cfg["S"], K=cfg["K"], G=64, amp=cfg["amp"], W=4)
    _, _, _, p1 = _banks(case, cfg["L"], cfg["B"])
    _, _, _, p2 = _banks(case, cfg["L"], cfg["B"], {"DPE_BCS_NO_CHIP2": "1"})
    _, _, _, p0 = _banks(case, cfg["L"], cfg["B"], {"DPE_BCS_NO_CHIP": "1"})
    print("bank ms: chip2 %.4f, chip %.4f, per-sample %.4f" % (p1["bcs_bank"][0], p2["bcs_bank"][0], p0["bcs_bank"][0]))
    assert p1["kernel"] == "bcs_bank_chip2_kernel" and p2["kernel"] == "bcs_bank_chip_kernel"     # 24.4 samples per chip: lanes <-> chips
    assert p1["bcs_bank"][0] < 0.8 * p0["bcs_bank"][0]
    worst = _check(case, cfg["L"], cfg["B"])
    print("config H (4 windows) worst rel err", worst)


def test_chip_kernel_lag_windows_wider_than_31_samples():
    """L = 100 at 25 Msps: chunks of 64 lags, 64 apart, each produced against the replica shifted by the chunk offset."""
    worst = _check(helpers.make_case(seed=15, fs=25e6, S=125000, K=4, G=64, amp=200.0), 100, 16)
    print("worst rel err", worst)


def test_chip_kernel_large_dc_offset():
    """(+900, -700) LSB of DC on a 40 LSB signal: the prefix sums then grow with the DC (|Q| ~ DC x 1088 per pass), which
    costs fp32 digits in the chip differences -- still inside the tolerance (1.4e-6 measured)."""
    c = helpers.make_case(seed=16, fs=25e6, S=250000, K=6, G=64, amp=40.0)
    for w in c["wins"]:
        iq = w["iq"].astype(np.int32)
        iq[0::2] += 900
        iq[1::2] -= 700
        w["iq"] = np.clip(iq, -32768, 32767).astype(np.int16)
    worst = _check(c, 31, 16)
    print("worst rel err", worst)


def test_chip_kernel_every_pass_per_block_count_gives_the_same_banks():
    """The moment block is the wave's tile of tpb passes (finalize is told its length): tpb = 1, 3 and the default must
    agree to fp32 rounding, nav-bit boundary inside a tile included."""
    case = helpers.make_case(seed=18, fs=25e6, S=250000, K=3, G=64, amp=80.0, flips=[True, False, True])
    ref_code, ref_carr, _, _ = _banks(case, 31, 16, {"DPE_BCS_NO_CHIP2": "1"})
    for tpb in ("1", "3", "7"):
        code, carr, _, _ = _banks(case, 31, 16, {"DPE_BCS_CHIP_TPB": tpb, "DPE_BCS_NO_CHIP2": "1"})
        for k in range(3):
            assert np.abs(code[0][k] - ref_code[0][k]).max() < TOL * np.abs(ref_code[0][k]).max()
            assert np.abs(carr[0][k] - ref_carr[0][k]).max() < TOL * np.abs(ref_carr[0][k]).max()
    ref_code, ref_carr, _, _ = _banks(case, 31, 16)
    for tpb in ("1", "3", "7"):
        code, carr, _, _ = _banks(case, 31, 16, {"DPE_BCS_CHIP_TPB": tpb, "DPE_BCS_CHIP2_P": tpb})
        for k in range(3):
            assert np.abs(code[0][k] - ref_code[0][k]).max() < TOL * np.abs(ref_code[0][k]).max()
            assert np.abs(carr[0][k] - ref_carr[0][k]).max() < TOL * np.abs(ref_carr[0][k]).max()


def test_chip_kernel_falls_back_when_not_eligible():
    """Carrier frequencies beyond the closed-form DC term's range (2 pi |fi| / fc > 0.25) take the per-sample kernels."""
    case = helpers.make_case(seed=19, fs=25e6, S=125000, K=2, G=64, amp=100.0)
    for w in case["wins"]:
        w["start"]["fi"] = w["start"]["fi"] + 60e3      # 60 kHz intermediate frequency
        ch = dict(w["start"])
        w["iq"] = dpe.synth.gen_iq(5, case["fs"], case["S"], ch, amp=100.0, flip=np.zeros(2, dtype=bool))
    _check(case, 31, 16)


N_C2 = int(os.environ.get("DPE_FUZZ_CHIP2_CASES", "30"))


@pytest.mark.parametrize("i", range(N_C2))
def test_random_case_through_the_second_form(i):
    """Seeded sweep aimed at bcs_bank_chip2_kernel: 20 / 25 Msps (19.5 / 24.4 samples per chip) and, every other case, one of the
    integer-nanosecond rates that give 16 ... 23 samples per chip (the kernel is instantiated for every chip length 16 ... 24;
    front ends at 16.368 / 20.46 / 24.552 Msps have non-integer-ns periods, i.e. the reference's ns-rounded time table, and
    stay on the per-sample kernels), window lengths that are
    multiples of nothing (so the clipped first and last chips, the circular wrap inside the margins and partial last
    passes of a tile all vary), 1 ... 12 SVs, 1 ... 3 windows, nav-bit boundary anywhere, lag windows 17 ... 31.  Banks of
    the second form, of the first form and of the per-sample kernels against the oracle; the second form must be the one
    that ran."""
    rng = np.random.Generator(np.random.PCG64(424243 + i))
    fs = float(rng.choice([20e6, 25e6]))
    if i % 2 == 1:   # sampling periods of 61, 58, 56, 53, 48, 46, 44, 42, 41 ns: floor(fs / fc) = 16, 16, 17, 18, 20, 21, 22, 23, 23
        fs = 1e9 / float(rng.choice([61, 58, 56, 53, 48, 46, 44, 42, 41]))
    S = 2 * int(rng.integers(3000, 90001))
    K = int(rng.choice([1, 2, 3, 5, 8, 12]))
    W = int(rng.choice([1, 1, 2, 3]))
    L = int(rng.integers(17, 32))
    C = 8 * (1 << int(np.ceil(np.log2(S))))
    b_max = int(np.floor((720 * 2e-7) ** (1.0 / 6.0) * C / (2 * np.pi * 127.5) * 0.999))
    # (the chip kernels drop a chip's second-order term: (2 pi B / C x chip length)^2 / 24 < 5e-7, checked at create)
    B = max(2, min(int(rng.choice([8, 12, 16, 24])), b_max, int(1.35e-4 * C / (2 * np.pi))))
    case = helpers.make_case(seed=7000 + i, fs=fs, S=S, K=K, G=16, amp=float(rng.choice([40.0, 100.0])), W=W)
    try:
        _, _, _, p = _banks(case, L, B)
        assert p["kernel"] == "bcs_bank_chip2_kernel"
        worst = _check(case, L, B)
    except Exception:
        print("chip2 sweep case %d: fs %.0f S %d K %d W %d L %d B %d" % (i, fs, S, K, W, L, B))
        raise
    print("case %d worst rel err %.3g" % (i, worst))


def test_device_ports_at_a_high_sampling_rate_take_the_chip_kernel():
    """dpe_bcs_update_dev at 25 Msps: the chip-boundary kernels are chosen by the channel values, which this form has on the
    device only -- it reads back the block its prep kernel derived (one small copy per window) instead of falling back
    to the per-sample kernels.  Same kernel and banks as the host form, both within the oracle's tolerance."""
    import torch
    from oracle import oracle as o
    L, B = 31, 16
    case = helpers.make_case(seed=23, fs=25e6, S=125000, K=4, G=64, amp=80.0)
    K, S, fs = case["K"], case["S"], case["fs"]
    iq, cs, _, _ = helpers.pack_gpu_inputs(case)
    d = torch.from_numpy(iq).to("cuda:0")
    s = case["wins"][0]["start"]

    def dv(a, dt):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=dt)).to("cuda:0")
    ports = dict(codePhaseStart=dv(s["rc"], np.float64), carrierPhaseStart=dv(s["ri"], np.float64), codeFrequency=dv(s["fc"], np.float64),
                 carrierFrequency=dv(s["fi"], np.float64), cpElapsedStart=dv(s["cp"], np.int32), cpReference=dv(s["cp_ref"], np.int32),
                 validPRNs=dv(s["prn"], np.uint8))
    out = {}
    for form in ("host", "dev", "hint"):
        bcs = dpe.BatchCorrScores(fs, samples_per_window=S, lag_half_width=L, bin_half_width=B, max_windows=1, max_channels=K)
        bcs.Start()
        if form == "host":
            bcs.Update(d, cs)
        else:
            if form == "hint":           # the caller's promise instead of the readback (dpe_bcs_set_dev_hint): same kernel, checked on the device
                bcs.set_dev_hint(1)
            bcs.UpdateDev(d, K, ports)
            assert bcs.dev_status() == 0
        out[form] = (bcs.read_banks(), bcs.read_info(), bcs.stage1_kernel)
        if form == "hint":               # a carrier offset beyond the promise is flagged (bit 3), the call still returns
            bad = dict(ports)
            bad["carrierFrequency"] = dv(np.asarray(s["fi"]) + 60e3, np.float64)
            bcs.UpdateDev(d, K, bad)
            assert bcs.dev_status() == 8 and bcs.stage1_kernel == "bcs_bank_chip2_kernel"
            # ... and the flagged window was RE-RUN on the device by the guarded per-sample kernels behind the chip kernel's launch:
            # its banks are the oracle's, like any other window's (the chip kernel alone is off by 1e-3 for this input)
            codeb, carrb = bcs.read_banks()
            for k in range(K):
                c, f, _ = o.bcs_sv(case["wins"][0]["iq"], fs, int(s["prn"][k]), s["rc"][k], s["ri"][k], s["fc"][k], s["fi"][k] + 60e3,
                                   int(s["cp"][k]), int(s["cp_ref"][k]), -L, L, -B, B, case["C"])
                assert np.abs(codeb[0][k] - c).max() < TOL * np.abs(c).max() and np.abs(carrb[0][k] - f).max() < TOL * np.abs(f).max()
            # ... and withdraws the hint: from the next call on the handle reads the derived block back again and chooses from the real
            # values -- the same out-of-promise input now takes a per-sample kernel, unflagged, and its banks are the oracle's
            bcs.UpdateDev(d, K, bad)
            assert bcs.dev_status() == 0 and bcs.stage1_kernel not in ("bcs_bank_chip2_kernel", "bcs_bank_chip_kernel")
            codeb, carrb = bcs.read_banks()
            for k in range(K):
                c, f, _ = o.bcs_sv(case["wins"][0]["iq"], fs, int(s["prn"][k]), s["rc"][k], s["ri"][k], s["fc"][k], s["fi"][k] + 60e3,
                                   int(s["cp"][k]), int(s["cp_ref"][k]), -L, L, -B, B, case["C"])
                assert np.abs(codeb[0][k] - c).max() < TOL * np.abs(c).max() and np.abs(carrb[0][k] - f).max() < TOL * np.abs(f).max()
            bcs.UpdateDev(d, K, ports)
            assert bcs.dev_status() == 0 and bcs.stage1_kernel == "bcs_bank_chip2_kernel"      # (chosen from the values read back)
        bcs.Stop()
    assert out["host"][2] == out["dev"][2] == out["hint"][2] == "bcs_bank_chip2_kernel"
    for a, b in zip(out["hint"][0], out["dev"][0]):       # nominal tile length: the banks of the two device forms agree to rounding
        assert np.abs(a - b).max() < 2e-7 * np.abs(b).max()
    (code0, carr0), info0, _ = out["host"]
    (code1, carr1), info1, _ = out["dev"]
    assert np.array_equal(info1[0], info0[0]) and np.array_equal(info1[1], info0[1]) and np.array_equal(info1[2], info0[2])
    for k in range(K):
        c, f, _ = o.bcs_sv(case["wins"][0]["iq"], fs, int(s["prn"][k]), s["rc"][k], s["ri"][k], s["fc"][k], s["fi"][k],
                           int(s["cp"][k]), int(s["cp_ref"][k]), -L, L, -B, B, case["C"])
        for got, ref in ((code1[0][k], c), (carr1[0][k], f)):
            assert np.abs(got - ref).max() < TOL * np.abs(ref).max()
        assert np.abs(code1[0][k] - code0[0][k]).max() < 2e-7 * np.abs(code0[0][k]).max()
        assert np.abs(carr1[0][k] - carr0[0][k]).max() < 2e-7 * np.abs(carr0[0][k]).max()


def test_sums_riding_in_the_chip2_launch_equal_the_sum_kernel():
    """Large batches through the second form (>= 48 windows by default, >= 8 here): the DC sums are computed by extra blocks of the chip2 launch itself
    (ride_sum_block, dpe_bcs_chip2.h) and handed to the correlator blocks through {epoch, I, Q} words; DPE_BCS_NO_SUMRIDE=1 keeps
    the DC-sum kernel in front.  Same integer sums => bit-identical means and banks; a window length with S % 4 = 2 (the last
    slot's tail; the padding between the windows must not be summed), a batch that grows between calls (slots the previous launch did not write are cleared), the wait never
    times out (device status 0)."""
    import torch
    from oracle import oracle as o
    case = helpers.make_case(seed=23, fs=25e6, S=100002, K=3, G=64, amp=60.0, W=9)
    iq, cs, _, _ = helpers.pack_gpu_inputs(case)
    stride = case["S"] + 2                  # windows 16-byte aligned (the riding form's condition) with S % 4 = 2 samples in each
    pad = np.full((9, 2 * stride), 1234, dtype=np.int16)
    pad[:, :2 * case["S"]] = iq
    d = torch.from_numpy(pad).to("cuda:0")
    out = {}
    # "forced": every correlator block behaves as if its wait had timed out and adds up its window's samples itself (DPE_BCS_RIDE_SPIN=-1;
    # "impatient": it gives up after the first poll) -- the fallback that keeps a broken dispatch-order assumption from ever producing
    # wrong means (batchcorrscores.cu:1065-1066: the mean is the exact sum over (float) S)
    for name, env in (("ride", {"DPE_BCS_SUMRIDE_MIN": "8"}), ("kernel", {"DPE_BCS_NO_SUMRIDE": "1"}),
                      ("forced", {"DPE_BCS_SUMRIDE_MIN": "8", "DPE_BCS_RIDE_SPIN": "-1"}), ("impatient", {"DPE_BCS_SUMRIDE_MIN": "8", "DPE_BCS_RIDE_SPIN": "0"})):   # (default: batches of >= 48 windows ride)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            bcs = dpe.BatchCorrScores(case["fs"], samples_per_window=case["S"], lag_half_width=31, bin_half_width=16, max_windows=9, max_channels=3)
            bcs.Start()
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        res = []
        for nw in (8, 9, 8, 9):          # 8 -> 9: the slot set grows; then epochs cycle
            bcs.Update(d, cs[:nw], window_stride=stride)
            assert bcs.stage1_kernel == "bcs_bank_chip2_kernel"
            code, carr = bcs.read_banks()
            res.append((code[:nw].copy(), carr[:nw].copy(), bcs.read_info()[2][:nw].copy()))
        for _ in range(20):              # more launches than there are epochs
            bcs.Update(d, cs, window_stride=stride)
        code, carr = bcs.read_banks()
        res.append((code.copy(), carr.copy(), bcs.read_info()[2].copy()))
        if name == "ride":
            assert bcs.dev_status() == 0          # no correlator block gave up waiting for its window's sums
        if name == "forced":
            assert bcs.dev_status() == 4 + 16     # this launch + sticky: the blocks summed their windows themselves
        bcs.Stop()
        out[name] = res
    for form in ("ride", "forced", "impatient"):
        for a, b in zip(out[form], out["kernel"]):
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), form
    for wi, w in enumerate(case["wins"]):
        s = w["start"]
        _, _, inf = o.bcs_sv(w["iq"], case["fs"], int(s["prn"][0]), s["rc"][0], s["ri"][0], s["fc"][0], s["fi"][0], int(s["cp"][0]),
                             int(s["cp_ref"][0]), -31, 31, -16, 16, case["C"])
        assert out["ride"][-1][2][wi] == inf["mean"]
