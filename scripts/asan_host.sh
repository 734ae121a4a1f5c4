#!/bin/bash
# Sanitizer pass over the host-side code (GPU ASAN is not available on the pool): the host-only sources of the C-ABI
# (cuChanMgr, EKF) built with g++ -fsanitize=address,undefined and driven by the CPU tests, and the C++ Module/Flow
# mirror's own test program.  Run from the repo root:   bash scripts/asan_host.sh
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/scratch
SAN="-O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
g++ -x c++ $SAN -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include \
    $ROOT/navlab-dpe-sdr_amd/csrc/dpe_chanmgr.hip $ROOT/navlab-dpe-sdr_amd/csrc/dpe_ekf.hip $ROOT/scripts/asan/stub.cpp \
    -o $ROOT/scratch/libhost_asan.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python3 $ROOT/scripts/asan/run.py
g++ $SAN $ROOT/navlab-dpe-sdr_amd/host/test_modules.cpp -I$ROOT/include -L$ROOT/navlab-dpe-sdr_amd -ldpe_hip \
    -Wl,-rpath,$ROOT/navlab-dpe-sdr_amd -lpthread -o $ROOT/scratch/test_modules_asan
$ROOT/scratch/test_modules_asan
echo "asan_host: clean"
