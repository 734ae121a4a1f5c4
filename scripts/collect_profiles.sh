#!/bin/bash
# Extra bench.py arguments (e.g. --config H) through DPE_BENCH_ARGS.
# Collects the judged rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo
# root):   bash scripts/collect_profiles.sh r1d
# Writes gpurun_out/<tag>_*: kernel stats, the bench line of the profiled run, an un-profiled bench line,
# and the FETCH_SIZE / WRITE_SIZE counter passes (separate runs, --kernel-trace only, as gpurun requires).
# Afterwards, in the repo:  python scripts/pmc_traffic.py gpurun_out <tag>  and copy the files to profiles/.
set -u
TAG=${1:-rX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --no-extras ${DPE_BENCH_ARGS:-} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --no-pipelined ${DPE_BENCH_ARGS:-} \
    > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err
cp $OUT/${TAG}_stats/*/*_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -- python3 $ROOT/bench.py --steps 3 --warmup 1 --clock-warmup-s 0 --batches 1 --min-batch-s 0 --no-extras --no-cpu-baseline --no-pipelined ${DPE_BENCH_ARGS:-} \
        > /dev/null 2> $OUT/${TAG}_pmc_$C.err
    cp $OUT/${TAG}_pmc_$C/*/*_counter_collection.csv $OUT/${TAG}_pmc_${C}_counter_collection.csv
done
ls -la $OUT | grep ${TAG}_
