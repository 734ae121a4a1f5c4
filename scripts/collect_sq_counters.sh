#!/bin/bash
# Extra bench.py arguments (e.g. --config H) through DPE_BENCH_ARGS.
# SQ counters per kernel for the default bench workload (run through gpurun from the repo root):
#   bash scripts/collect_sq_counters.sh r1e      ->  gpurun_out/<tag>_sq_*/...counter_collection.csv
# Four passes of four counters each, --kernel-trace only (gpurun refuses --pmc with the API trace domains).
# Summarise afterwards with:  python scripts/sq_summary.py gpurun_out <tag> <bench.json with the kernel times>
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
# (round 3: + the stall-attribution set -- SQ_WAIT_INST_LDS = LDS-issue stalls, a sub-bucket of SQ_WAIT_INST_ANY; SQ_WAIT_ANY =
#  parked at s_waitcnt; transcendental and fp64 instruction counts; scalar / misc issue activity)
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_FMA_F64" "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_sq_$i -- python3 $R/bench.py --steps 2 --warmup 1 --clock-warmup-s 0 --batches 1 --min-batch-s 0 --no-extras --no-cpu-baseline ${DPE_BENCH_ARGS:-} \
      > /dev/null 2> $R/gpurun_out/${TAG}_sq_$i.err
done
ls $R/gpurun_out | grep ${TAG}_sq_
