#!/bin/bash
# The long runs behind profiles/<tag>_robustness.txt, on the GPU box:  bash scripts/robustness.sh r6
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r6}
O=$R/gpurun_out/${T}_robustness.txt
cd $R
: > $O
run() { echo "\$ $*" >> $O; ( eval "$@" ) 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | tail -${TAILN:-2} >> $O; echo >> $O; }
run python -m pytest tests -m gpu -q
run DPE_FUZZ_CASES=1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q
run DPE_FUZZ_CASES=400 DPE_FUZZ_WIDE_CASES=100 DPE_FUZZ_ACQ_CASES=160 DPE_FUZZ_CHIP2_CASES=80 python -m pytest tests/test_gpu_fuzz.py -m gpu -q
run python scripts/stress_pipe.py 3000
run timeout 200 python scripts/stress_sum_ride.py
run python scripts/soak_device_loop.py
run python scripts/stress_polling.py 200000
run python -c "'import __graft_entry__ as g; g.smoke()'"
cat $O
