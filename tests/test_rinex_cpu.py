"""RINEX navigation reader: the reference's DPInit takes its ephemerides from a RINEX 2 nav file
(dpinit.cpp:130-144) and cuChanMgr picks the set closest in toe (cuchanmgr.cu:269-299).  Fixture O11 = an excerpt of the
reference's own data file demofiles/nist1860.18n + what the reference's Python parser (libgnss/rinex.py) returns."""
import os
import subprocess

import numpy as np

import navlab_dpe_sdr_amd as dpe
from navlab_dpe_sdr_amd import rinex

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXCERPT = os.path.join(HERE, "o11_nist1860_excerpt.18n")


def test_reader_matches_the_reference_parser(golden):
    g = golden("o11_rinex")
    nav = rinex.read_rinex_nav(EXCERPT)
    assert nav["prn"].size > 40 and set(nav["week"]) == {1869, 2008}      # one stale 2015 record, the rest July 2018
    for k, p in enumerate(g["prns"]):
        first = int(np.nonzero(nav["prn"] == p)[0][0])                     # libgnss/rinex.py: first record of the PRN
        assert np.array_equal(nav["eph"][first], g["eph_first"][k]), p


def test_selection_reproduces_the_handoff_ephemerides(golden):
    """Closest-toe choice at the handoff time = the ephemerides PyGNSS decoded from the signal and wrote into the
    handoff file (12 significant digits in the RINEX text)."""
    g = golden("o11_rinex")
    nav = rinex.read_rinex_nav(EXCERPT)
    sel = rinex.select_ephemerides(nav, g["handoff_prns"], float(g["handoff_rxTime"]))
    ref = g["handoff_eph"]
    assert np.abs(sel - ref).max(axis=0)[dpe.handoff.EPH_FIELDS.index("t_oe")] == 0
    assert (np.abs(sel - ref) <= 1e-11 * np.maximum(np.abs(ref), 1e-30)).all()
    # strictly-closer rule: midway between two sets the earlier-listed one stays
    toes = sorted(set(int(t) for t, p in zip(nav["eph"][:, 15], nav["prn"]) if p == 2 and t > 4e5))
    if len(toes) >= 2:
        mid = 0.5 * (toes[0] + toes[1])
        a = rinex.select_ephemerides(nav, [2], mid)[0, 15]
        order = [int(t) for t, p in zip(nav["eph"][:, 15], nav["prn"]) if p == 2 and int(t) in toes[:2]]
        assert a == order[0]


def test_cpp_reader_agrees_with_python(golden):
    g = golden("o11_rinex")
    exe = os.path.join(os.path.dirname(dpe.engine.LIB_PATH), "dpe_flow")
    prns = [int(p) for p in g["handoff_prns"]]
    t = float(g["handoff_rxTime"])
    out = subprocess.run([exe, "--dump-eph", EXCERPT, repr(t)] + [str(p) for p in prns], capture_output=True, text=True, check=True).stdout
    lines = out.strip().splitlines()
    nav = rinex.read_rinex_nav(EXCERPT)
    assert lines[0] == "%d records" % nav["prn"].size
    sel = rinex.select_ephemerides(nav, prns, t)
    for k, line in enumerate(lines[1:]):
        vals = [float(v) for v in line.split(",")]
        assert int(vals[0]) == prns[k] and np.array_equal(np.array(vals[1:]), sel[k])
    bad = subprocess.run([exe, "--dump-eph", EXCERPT, "0", "33"], capture_output=True, text=True)
    assert bad.returncode != 0 and "no ephemeris for PRN 33" in bad.stderr
