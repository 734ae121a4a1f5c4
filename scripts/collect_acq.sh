#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r6}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --acq coherent > $R/gpurun_out/${T}_acq_bench.json 2> $R/gpurun_out/${T}_acq_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_acq_stats -- python3 $R/bench.py --acq coherent > $R/gpurun_out/${T}_acq_bench_under_rocprof.json 2> $R/gpurun_out/${T}_acq_stats.err
cp $R/gpurun_out/${T}_acq_stats/*/*_kernel_stats.csv $R/gpurun_out/${T}_acq_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_acqnc_stats -- python3 $R/scripts/acq/one_noncoherent.py 40 > /dev/null 2> $R/gpurun_out/${T}_acqnc_stats.err
cp $R/gpurun_out/${T}_acqnc_stats/*/*_kernel_stats.csv $R/gpurun_out/${T}_acq_noncoherent_kernel_stats.csv
rm -rf $R/gpurun_out/${T}_acq_stats $R/gpurun_out/${T}_acqnc_stats
