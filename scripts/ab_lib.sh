#!/bin/bash
# A/B of library builds on one box: alternating runs of `bench.py --config $CFG` with DPE_LIB_PATH pointing at each build kept under
# scratch/ab/<name>/libdpe_hip.so, and the in-tree build ("tree").  Usage: CFG=H ROUNDS=3 bash scripts/ab_lib.sh base tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
CFG=${CFG:-H}
ROUNDS=${ROUNDS:-3}
OUT=$R/gpurun_out/ab_${CFG}_$(echo "$@" | tr ' ' '_').txt
mkdir -p $R/gpurun_out
: > $OUT
for r in $(seq $ROUNDS); do
  for b in "$@"; do
    if [ "$b" = tree ]; then L=$R/navlab-dpe-sdr_amd/libdpe_hip.so; else L=$R/scratch/ab/$b/libdpe_hip.so; fi
    DPE_LIB_PATH=$L python3 bench.py --config $CFG --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('round $r $b', 'ms_per_step %.4f' % d['ms_per_step'], 'one_stream %.4f' % d.get('one_stream_ms_per_step', 0), 'kernel', d['roofline']['kernel'], '%.4f' % d['roofline']['avg_launch_ms'], 'status', d.get('stage1_dev_status'), {k: round(v, 4) for k, v in d['kernels_ms_per_step'].items()})
" | tee -a $OUT
  done
done
