#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
W, fs, S, K = 400, 2.5e6, 50000, 8
iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0)
os.makedirs("/tmp/lt", exist_ok=True)
iq.tofile("/tmp/lt/s.dat")
with open(dpe.workload.HANDOFF_CSV) as f, open("/tmp/lt/handoff.csv", "w") as g:
    for line in f:
        g.write("bytes_read,0\n" if line.startswith("bytes_read") else line)
PY
cd /tmp && export TMPDIR=/tmp
EXE=${DPE_FLOW_EXE:-$R/navlab-dpe-sdr_amd/dpe_flow}
for V in "" "--ekf"; do
  T=lt_pass; [ -n "$V" ] && T=lt_ekf
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T -- $EXE --samples /tmp/lt/s.dat --handoff /tmp/lt/handoff.csv --out /tmp/lt/X.csv --iters 400 --grid-dim 25 --spacing 1.0 --device-loop $V > /dev/null 2> $R/gpurun_out/$T.err
  echo "== $T"; grep "second half" $R/gpurun_out/$T.err
  python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/$T/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print("  ", r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3, 2))
PY
done
