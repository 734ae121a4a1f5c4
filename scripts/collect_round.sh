#!/bin/bash
# Round evidence (tag = first argument, default r6): scripts/collect_profiles.sh for R / H / M, the acquisition line under rocprofv3, SQ counters for R and H.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r6}
cd $R
bash scripts/collect_profiles.sh ${T}_R > /dev/null 2>&1
DPE_BENCH_ARGS="--config H" bash scripts/collect_profiles.sh ${T}_H > /dev/null 2>&1
DPE_BENCH_ARGS="--config M" bash scripts/collect_profiles.sh ${T}_M > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --acq coherent > $R/gpurun_out/${T}_acq_bench.json 2> $R/gpurun_out/${T}_acq_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_acq_stats -- python3 $R/bench.py --acq coherent > $R/gpurun_out/${T}_acq_bench_under_rocprof.json 2> $R/gpurun_out/${T}_acq_stats.err
cp $R/gpurun_out/${T}_acq_stats/*/*_kernel_stats.csv $R/gpurun_out/${T}_acq_kernel_stats.csv
cd $R
bash scripts/collect_sq_counters.sh ${T}_R > /dev/null 2>&1
DPE_BENCH_ARGS="--config H" bash scripts/collect_sq_counters.sh ${T}_H > /dev/null 2>&1
python3 scripts/pmc_traffic.py gpurun_out ${T}_R 256 R
python3 scripts/pmc_traffic.py gpurun_out ${T}_H 128 H
python3 scripts/pmc_traffic.py gpurun_out ${T}_M 256 M
python3 scripts/sq_summary.py gpurun_out ${T}_R gpurun_out/${T}_R_bench.json
python3 scripts/sq_summary.py gpurun_out ${T}_H gpurun_out/${T}_H_bench.json
# keep the merge small: the raw traces are not needed, only the summaries
rm -rf gpurun_out/${T}_*_stats gpurun_out/${T}_*_pmc_FETCH_SIZE gpurun_out/${T}_*_pmc_WRITE_SIZE gpurun_out/${T}_*_sq_[0-9]
du -sh gpurun_out
