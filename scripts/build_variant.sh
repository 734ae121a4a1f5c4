#!/bin/bash
# An A/B build of the library beside the in-tree one: recompiles the listed translation units with extra flags and links them with
# the in-tree objects of the others (navlab-dpe-sdr_amd/build/*.o from __graft_entry__.build()) into scratch/ab/<name>/libdpe_hip.so.
#   scripts/build_variant.sh <name> "<extra hipcc flags>" dpe_bcs [dpe_acq ...]
# Run one with DPE_LIB_PATH=scratch/ab/<name>/libdpe_hip.so (scripts/ab_lib.sh does).
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; flags=$2; shift 2
out=$R/scratch/ab/$name
mkdir -p $out
cd $R/navlab-dpe-sdr_amd
objs=""
for o in build/*.o; do
  b=$(basename $o .o)
  use=$o
  for u in "$@"; do
    if [ "$u" = "$b" ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed $flags -c csrc/$b.hip -o $out/$b.o &
      use=$out/$b.o
    fi
  done
  objs="$objs $use"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -lrocfft -ldl -o $out/libdpe_hip.so
echo built $out/libdpe_hip.so
