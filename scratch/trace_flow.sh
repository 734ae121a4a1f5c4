#!/bin/bash
# builds a sample file, then traces the C++ flow
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
W, fs, S, K = 200, 2.5e6, 50000, 8
iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0)
iq.tofile("/tmp/s.dat")
with open(dpe.workload.HANDOFF_CSV) as f, open("/tmp/handoff.csv", "w") as g:
    for line in f:
        g.write("bytes_read,0\n" if line.startswith("bytes_read") else line)
PY
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d gpurun_out/trace_flow -- ./navlab-dpe-sdr_amd/dpe_flow --samples /tmp/s.dat --handoff /tmp/handoff.csv --out /tmp/X.csv --iters 200 --grid-dim 25 --spacing 1.0 --no-graph 2>&1 | tail -3
ls -R gpurun_out/trace_flow | head
