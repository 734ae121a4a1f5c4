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
    assert built.dpe_abi_version() == 1


def test_ca_code_host(built, golden):
    assert np.array_equal(dpe.engine.gen_ca_code(), golden("o1_ca_chips")["chips"])


def test_struct_layouts_match_header():
    import ctypes as C
    e = dpe.engine
    assert C.sizeof(e.BcsConfig) == 32 and C.sizeof(e.ChanStart) == 48
    assert C.sizeof(e.BcmWindow) == 152 and C.sizeof(e.ChanEnd) == 104
    assert C.sizeof(e.BcmResult) == 104 and C.sizeof(e.BcmConfig) == 96


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    bcs = dpe.BatchCorrScores(2.5e6, samples_per_window=50000)
    with pytest.raises(dpe.DpeError):
        bcs.Start()          # hipMalloc fails -> error, never a CPU path
    with pytest.raises(dpe.DpeError):
        bcs.Update(0, dpe.engine.chan_start_array([2], [0.0], [0.0], [1.023e6], [0.0], [0], [0]))
