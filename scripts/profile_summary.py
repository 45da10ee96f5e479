"""Condense a scripts/profile_round.sh output directory into the small files that get committed
under profiles/: per-kernel duration stats + HBM traffic per launch from the PMC passes,
corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes and calibrated on a
known 1-GiB read in the same access pattern."""
import csv, json, os, sys
from collections import defaultdict
d = sys.argv[1]

def counters(path):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        acc[(row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

out = {"kernels": []}
for row in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
    if "tilespmv" in row["Name"]:
        out["kernels"].append({"name": row["Name"].split("(")[0].replace("void ", ""), "calls": int(row["Calls"]),
                               "avg_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])})
fetch = counters(os.path.join(d, "pmc_FETCH_SIZE.csv")); write = counters(os.path.join(d, "pmc_WRITE_SIZE.csv"))
cal = counters(os.path.join(d, "calib_FETCH_SIZE.csv"))
calib = {k[0]: v for k, v in cal.items()}
# every microbenchmark kernel reads exactly 1 GiB = 1048576 KB
ratios = {k: 1048576.0 / v for k, v in calib.items() if v > 0}
grp = [v for k, v in ratios.items() if "k_group_strips" in k]
factor = sum(grp) / len(grp) if grp else 2.0
out["fetch_size_calibration"] = {"known_read_kb": 1048576, "reported_kb_by_kernel": calib, "true_over_reported_group_strips_8B_per_lane": factor}
tr = {}
for (k, c), v in fetch.items():
    if "tilespmv" in k: tr.setdefault(k, {})["FETCH_SIZE_kb"] = v
for (k, c), v in write.items():
    if "tilespmv" in k: tr.setdefault(k, {})["WRITE_SIZE_kb"] = v
for k, t in tr.items():
    t["hbm_bytes_per_launch"] = int((t.get("FETCH_SIZE_kb", 0) * factor + t.get("WRITE_SIZE_kb", 0)) * 1024)
out["traffic"] = tr
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
