"""Runs the host-side CPU tests (cuChanMgr, EKF, RINEX) against the ASAN/UBSAN build of the host-only sources.
Started by scripts/asan_host.sh with libasan / libubsan preloaded."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import navlab_dpe_sdr_amd as dpe  # noqa: E402
import pytest  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "scratch", "libhost_asan.so"))
lib.dpe_last_error.restype = C.c_char_p
dpe.engine._lib = lib          # the sanitised library stands in for libdpe_hip.so (host entry points only)
os.chdir(ROOT)
rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "tests/test_abi_cpu.py", "-k", "chanmgr"])
rc = rc or pytest.main(["-x", "-q", "-p", "no:cacheprovider", "tests/test_ekf_cpu.py", "tests/test_rinex_cpu.py"])
sys.exit(rc)
