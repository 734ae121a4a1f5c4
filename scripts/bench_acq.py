#!/usr/bin/env python3
"""Coarse-acquisition timing: thin wrapper of `bench.py --acq <coherent|noncoherent|textbook>` (kept for the command lines
quoted in DESIGN.md)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--acq", sys.argv[1] if len(sys.argv) > 1 else "coherent"]))
