"""CPU: the C-ABI library loads and exports every symbol include/dpe_hip.h declares; host-only
entry points work without a GPU; product paths fail loudly without one."""
import os
import re

import numpy as np
import pytest

import navlab_dpe_sdr_amd as dpe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    return dpe.engine.lib()


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "dpe_hip.h")).read()
    names = set(re.findall(r"\b(dpe_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(built, n), n
    assert names == set(dpe.engine.EXPORTS)
    assert built.dpe_abi_version() == 4


def test_integration_md_has_a_row_for_every_exported_symbol():
    """INTEGRATION.md section 1b names, for every function include/dpe_hip.h declares, the reference interface it replaces (file:line)
    or says why there is none -- one row per symbol, none missing, none stale."""
    hdr = open(os.path.join(ROOT, "include", "dpe_hip.h")).read()
    names = set(re.findall(r"\b(dpe_[a-z0-9_]+)\s*\(", hdr))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 1b."):]
    sec = sec[:sec.index("\nError convention")]
    rows = dict(re.findall(r"^\| `(dpe_[a-z0-9_]+)` \| (.+) \|$", sec, flags=re.M))
    assert set(rows) == names, (sorted(names - set(rows)), sorted(set(rows) - names))
    assert all(len(v) > 3 and v != "?" for v in rows.values())


def test_ca_code_host(built, golden):
    assert np.array_equal(dpe.engine.gen_ca_code(), golden("o1_ca_chips")["chips"])


def test_struct_layouts_match_header():
    import ctypes as C
    e = dpe.engine
    assert C.sizeof(e.BcsConfig) == 32 and C.sizeof(e.ChanStart) == 48
    assert C.sizeof(e.BcmWindow) == 152 and C.sizeof(e.ChanEnd) == 104
    assert C.sizeof(e.BcmResult) == 248 and C.sizeof(e.BcmConfig) == 104


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    bcs = dpe.BatchCorrScores(2.5e6, samples_per_window=50000)
    with pytest.raises(dpe.DpeError):
        bcs.Start()          # hipMalloc fails -> error, never a CPU path
    with pytest.raises(dpe.DpeError):
        bcs.Update(0, dpe.engine.chan_start_array([2], [0.0], [0.0], [1.023e6], [0.0], [0], [0]))
    g = dpe.synth.rand_grid(3, 64)
    with pytest.raises(dpe.DpeError):      # the batches-in-flight form creates its lanes' handles at once: the same loud failure
        dpe.Pipe(2.5e6, 50000, g, g, lag_half_width=4, bin_half_width=20, max_windows=2, max_channels=8, in_flight=2)


def test_chanmgr_matches_oracle(built, oracle):
    """Product cuChanMgr (host C++ in libdpe_hip.so) vs the oracle's independent restatement:
    Start + 3 Updates on the handoff state, spread time grid."""
    from tests import helpers
    ho = dpe.handoff.read_handoff(helpers.HANDOFF)
    T = 0.02
    X = ho["X_ECEF"]
    pos, _ = dpe.synth.spread_grid()
    tg = np.unique(pos[:, 3])
    cm = dpe.ChanMgr.from_handoff(ho, T)
    om = oracle.ChanMgr(ho["prn_list"], ho["rc"], ho["ri"], ho["fc"], ho["fi"], ho["cp"], ho["cp_timestamp"],
                        ho["TOW"], ho["eph"], ho["rxTime"], T)
    for it in range(4):
        xk = X + np.array([0.3, -0.2, 0.1, 0.5, 0.01, 0.0, -0.02, 0.003]) * it   # a moving fix
        if it == 0:
            cm.Start(xk, xk, tg)
            ob, oR = om.start(xk, xk, tg)
        else:
            cm.Update(xk, xk, tg)
            ob, oR = om.update(xk, xk, tg)
        s, e, w, batch = cm.outputs(with_batch=True)
        assert np.array_equal(s["codePhaseStart"], om.rcStart) and np.array_equal(s["cpElapsedStart"], om.cpElaStart)
        assert np.abs(e["codePhaseEnd"] - om.rcEnd).max() < 1e-9 and np.array_equal(e["cpElapsedEnd"], om.cpElaEnd)
        assert np.abs(s["carrierPhaseStart"] - om.riStart).max() < 1e-12
        assert np.abs(s["codeFrequency"] - om.fc).max() < 1e-6 and np.abs(s["carrierFrequency"] - om.fi).max() < 1e-6
        assert w["rxTime"][0] == om.rxTime
        assert np.abs(batch - ob).max() < 1e-6 and np.abs(w["enu2ecef"][0] - oR).max() < 1e-14
        assert np.abs(e["satState"] - ob[:, tg.size // 2]).max() < 1e-6
    cm.Stop()


@pytest.mark.parametrize("seed,T,n", [(1, 0.02, 120), (2, 0.005, 150), (3, 0.001, 200), (4, 0.02, 60)])
def test_chanmgr_long_runs_random_subsets(built, oracle, seed, T, n):
    """Product cuChanMgr vs the oracle over long runs: random SV subsets, window lengths of 1 / 5 / 20 ms (code-period
    counters and nav-bit references roll over many times), a receiver that accelerates and whose fix is noisy.
    Integer state must agree exactly at every step; phases must not drift apart."""
    from tests import helpers
    rng = np.random.Generator(np.random.PCG64(seed))
    ho = dpe.handoff.read_handoff(helpers.HANDOFF)
    K = int(rng.integers(1, 9))
    sel = np.sort(rng.choice(8, size=K, replace=False))
    sub = dict(ho)
    for key in ("prn_list", "rc", "ri", "fc", "fi", "cp", "cp_timestamp", "TOW", "eph"):
        sub[key] = ho[key][sel]
    tg = np.array([-2.0, -1.0, 0.0, 1.0, 2.0]) * 6.0
    cm = dpe.ChanMgr.from_handoff(sub, T)
    om = oracle.ChanMgr(sub["prn_list"], sub["rc"], sub["ri"], sub["fc"], sub["fi"], sub["cp"], sub["cp_timestamp"],
                        sub["TOW"], sub["eph"], sub["rxTime"], T)
    x = ho["X_ECEF"].copy()
    v = rng.uniform(-30.0, 30.0, 3)
    for it in range(n):
        x[:3] += v * T
        v += rng.uniform(-2.0, 2.0, 3) * T
        x[4:7] = v
        xk = x + np.concatenate([rng.normal(0, 2.0, 4), rng.normal(0, 0.1, 4)])       # the fix fed back
        centre = xk + np.concatenate([rng.normal(0, 1.0, 4), np.zeros(4)])
        if it == 0:
            cm.Start(xk, centre, tg)
            ob, oR = om.start(xk, centre, tg)
        else:
            cm.Update(xk, centre, tg)
            ob, oR = om.update(xk, centre, tg)
        s, e, w, batch = cm.outputs(with_batch=True)
        assert np.array_equal(s["cpElapsedStart"], om.cpElaStart) and np.array_equal(e["cpElapsedEnd"], om.cpElaEnd), it
        assert np.array_equal(e["cpRef"], om.cpRef) and np.array_equal(e["cpRefTOW"], om.cpRefTOW), it
        assert np.abs(s["codePhaseStart"] - om.rcStart).max() < 1e-8 and np.abs(e["codePhaseEnd"] - om.rcEnd).max() < 1e-8, it
        d = np.abs(s["carrierPhaseStart"] - om.riStart)
        assert np.minimum(d, 1.0 - d).max() < 1e-7, it
        assert np.abs(s["codeFrequency"] - om.fc).max() < 1e-6 and np.abs(s["carrierFrequency"] - om.fi).max() < 1e-6, it
        assert w["rxTime"][0] == om.rxTime, it
        assert np.abs(batch - ob).max() < 1e-5 and np.abs(w["enu2ecef"][0] - oR).max() < 1e-13, it
    cm.Stop()


@pytest.mark.parametrize("gtype,dim,sp", [(0, 5, 1.0), (0, 6, 0.5), (2, 9, 1.5), (2, 25, 1.0)])
def test_host_grid_builders_match_oracle(built, oracle, tmp_path, gtype, dim, sp):
    """C++ host (host/grids.hpp via `dpe_flow --dump-grid`) vs the oracle's BCM_InitPosGrid restatement."""
    import subprocess
    out = str(tmp_path / "g.bin")
    subprocess.check_call([os.path.join(ROOT, "navlab-dpe-sdr_amd", "dpe_flow"), "--dump-grid", str(gtype), str(dim), str(sp), out])
    raw = np.fromfile(out)
    G = dim ** 4
    og, otg = oracle.init_grid(gtype, [dim] * 4, [sp] * 4)
    assert np.array_equal(raw[:4 * G].reshape(G, 4), og) and np.array_equal(raw[4 * G:], otg)
    if gtype == 0:
        assert np.array_equal(og, dpe.synth.uniform_grid(dim, sp))


def test_host_module_and_flow_interface(built):
    """C++ mirror of dsp::Module / dsp::Flow (host/dsp.hpp, host/modules.hpp): parameter typing, port validation by
    ValueType + VectorLength, wiring by names, Start roll-back, Update-before-Start -- the reference's rules
    (module.cpp:21-63,284-320; flow.cu:28-87,212-324).  host/test_modules.cpp, no GPU work."""
    import subprocess
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "test_modules")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
