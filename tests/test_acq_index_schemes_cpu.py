"""CPU: the index schemes of the hand-written acquisition transforms, restated in numpy (scripts/proto/) and checked against
numpy.fft in fp64 -- the prime-factor 50-point transform, the 50 x 50 two-step 2 500-point transform and the decimation in time by
ten of csrc/dpe_acq_pack.h; the generic four-pass in-place mixed-radix transform of csrc/dpe_acq_mixed.h."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "proto", script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_packed_transform_index_scheme():
    out = _run("acq_pack_proto.py")
    errs = [float(x) for x in re.findall(r"err ([0-9.e+-]+)", out)]
    assert len(errs) == 12 and max(errs) < 1e-11, out      # idft50, ten residue classes, the alias-summed surface


def test_mixed_radix_index_scheme():
    out = _run("acq_mixed_radix_proto.py")
    rows = [l.split() for l in out.strip().splitlines()]
    assert [r[0] for r in rows] == ["idft8", "4000", "5000", "2500"] and max(float(r[1]) for r in rows) < 1e-11, out
