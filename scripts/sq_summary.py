"""Summarises scripts/collect_sq_counters.sh: mean counter value per launch and kernel, plus the VALU issue utilisation
SQ_INSTS_VALU / 1024 SIMDs x 4 cycles / (kernel time x 2.4 GHz) with the kernel times of a bench line.
usage: python scripts/sq_summary.py <dir> <tag> <bench.json>  ->  <dir>/<tag>_sq_counters.json"""
import csv, glob, json, re, sys
from collections import defaultdict

d, tag, bench = sys.argv[1], sys.argv[2], json.load(open(sys.argv[3]))
t = bench["kernels_ms_per_step"]
times = {"bcm_scan_kernel": t["bcm_scan"], "bcs_bank": t["bcs_bank"], "bcs_finalize_kernel": t["bcs_finalize"], "bcs_sum_kernel": t["bcs_sum"]}
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{d}/{tag}_sq_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"])).replace("dpe::", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"source": "scripts/collect_sq_counters.sh (rocprofv3 --pmc, 4 passes x 4 counters, --kernel-trace only); mean per launch, whole device",
       "kernels": {}}
for k, v in acc.items():
    if k.startswith("__amd"):
        continue
    e = {c: sum(x) / len(x) for c, x in v.items()}
    for name, ms in times.items():
        if k.startswith(name) and "SQ_INSTS_VALU" in e:
            e["valu_issue_utilisation"] = e["SQ_INSTS_VALU"] / 1024 * 4 / (ms * 1e-3 * 2.4e9)
            e["kernel_ms"] = ms
            if "SQ_LDS_IDX_ACTIVE" in e:   # LDS-array cycles over the 256 CUs x kernel cycles
                e["lds_array_utilisation"] = e["SQ_LDS_IDX_ACTIVE"] / 256 / (ms * 1e-3 * 2.4e9)
    out["kernels"][k] = e
json.dump(out, open(f"{d}/{tag}_sq_counters.json", "w"), indent=1)
for k, e in out["kernels"].items():
    print(k[:48], "VALU", int(e.get("SQ_INSTS_VALU", 0)), "util", round(e.get("valu_issue_utilisation", 0), 3))
