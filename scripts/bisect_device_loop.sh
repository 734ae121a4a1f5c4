#!/bin/bash
# Bisect of the device-resident loop's per-window time over library builds kept under scratch/bisect/<sha>/ (libdpe_hip.so + dpe_flow,
# built by /tmp/bis/build_one.sh from `git worktree` checkouts) and the in-tree build (HEAD): dpe_flow --device-loop, 25^4 grids,
# DPE_LAT_WINDOWS windows (default 2000), the builds in alternating order, ROUNDS times.  Output: gpurun_out/bisect_device_loop.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
W=${DPE_LAT_WINDOWS:-2000}
ROUNDS=${ROUNDS:-3}
OUT=$R/gpurun_out/bisect_device_loop.txt
mkdir -p $R/gpurun_out
python3 - <<PY
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import navlab_dpe_sdr_amd as dpe
W, fs, S, K = $W, 2.5e6, 50000, 8
iq, _, _, _ = dpe.workload.build_windows(W, fs, S, K, seed=5, amp=200.0)
os.makedirs("/tmp/lt", exist_ok=True)
iq.tofile("/tmp/lt/s.dat")
with open(dpe.workload.HANDOFF_CSV) as f, open("/tmp/lt/handoff.csv", "w") as g:
    for line in f:
        g.write("bytes_read,0\n" if line.startswith("bytes_read") else line)
PY
: > $OUT
BUILDS="$(ls -d scratch/bisect/*/ 2>/dev/null | tr '\n' ' ') navlab-dpe-sdr_amd/"
for r in $(seq $ROUNDS); do
  for b in $BUILDS; do
    for V in "--device-loop" ""; do
      L=$( $b/dpe_flow --samples /tmp/lt/s.dat --handoff /tmp/lt/handoff.csv --out /tmp/lt/X.csv --iters $W --grid-dim 25 --spacing 1.0 $V 2>&1 >/dev/null | grep "second half" )
      echo "round $r $b [$V] $L" | tee -a $OUT
    done
  done
done
