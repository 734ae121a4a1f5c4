"""Builds profiles/<tag>_pmc_traffic.json from the two rocprofv3 counter passes of scripts/collect_profiles.sh:
mean FETCH_SIZE / WRITE_SIZE (KB) per launch and kernel, HBM bytes = (2 x FETCH + WRITE) x 1024 -- gfx950
tallies 128-byte fetches as 64 B (MI355X_MICROARCH.md), cross-checked on bcs_sum_kernel, whose only traffic
is the sample read (windows x S x 4 bytes).   usage: python scripts/pmc_traffic.py <dir> <tag> [windows] [config]"""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    d, tag = sys.argv[1], sys.argv[2]
    windows = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    config = sys.argv[4] if len(sys.argv) > 4 else "R"
    acc = defaultdict(lambda: defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        with open(f"{d}/{tag}_pmc_{c}_counter_collection.csv") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == c:
                    acc[short(row["Kernel_Name"])][(c, row["Dispatch_Id"])].append(float(row["Counter_Value"]))
    out = {"source": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- "
                     "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline [--config C]   (scripts/collect_profiles.sh)",
           "note": "KB per launch (mean over launches, summed over the counter's instances); FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md; bcs_sum_kernel cross-check: 2*FETCH = windows*S*4 bytes",
           "windows_per_step": windows, "config": config, "kernels": {}}
    for k, v in acc.items():
        per = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            launches = [sum(vals) for (cc, _), vals in v.items() if cc == c]
            per[c] = sum(launches) / len(launches) if launches else 0.0
        key = "bcm_scan_kernel" if "bcm_scan_kernel" in k else k
        out["kernels"][key] = {"full_name": k, "FETCH_SIZE_KB": per["FETCH_SIZE"], "WRITE_SIZE_KB": per["WRITE_SIZE"],
                               "hbm_bytes_per_launch": (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0}
    json.dump(out, open(f"{d}/{tag}_pmc_traffic.json", "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:40s} fetch {v['FETCH_SIZE_KB']:12.1f} KB  write {v['WRITE_SIZE_KB']:12.1f} KB  hbm {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB")


if __name__ == "__main__":
    main()
