#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/acqt_kt -- python3 $R/scripts/acq/one_mode.py 20 > /dev/null 2> $R/gpurun_out/acqt_kt.err
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/acqt_sq_$i -- python3 $R/scripts/acq/one_mode.py 3 > /dev/null 2> $R/gpurun_out/acqt_sq_$i.err
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/acqt_kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:5]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/acqt_sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "acq_corr2500" not in k: continue
    print(k)
    for c, v in sorted(d.items()): print("   %-28s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf gpurun_out/acqt_sq_* gpurun_out/acqt_kt
