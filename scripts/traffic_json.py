"""Condense one scripts/profile_traffic.sh directory into traffic_<workload>_<dtype>.json (+ a short kernel table):
FETCH_SIZE / WRITE_SIZE per launch of the dominant tilespmv kernel, FETCH_SIZE corrected by the factor measured on a
known 1-GiB read in the same session (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950)."""
import csv, json, os, sys, time
from collections import defaultdict
d, wl, dt, calib = sys.argv[1:5]

def counters(path):
    acc = defaultdict(list)
    for row in csv.DictReader(open(path)):
        acc[(row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

kern = []
for row in csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))):
    if "tilespmv" in row["Name"]:
        kern.append({"name": row["Name"].split("(")[0].replace("void ", ""), "calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                     "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]), "percent": float(row["Percentage"])})
kern.sort(key=lambda k: -k["avg_ns"] * k["calls"])
factor = 2.0
try:
    cal = counters(calib)
    r = [1048576.0 / v[0] for k, v in cal.items() if "k_group_strips" in k[0] and v[0] > 0]
    if r:
        factor = sum(r) / len(r)
except Exception as e:
    print("no calibration:", e)
fetch = counters(os.path.join(d, "pmc_FETCH_SIZE.csv")); write = counters(os.path.join(d, "pmc_WRITE_SIZE.csv"))
dom = kern[0]["name"] if kern else None
per = {}
for (k, c), (v, n) in list(fetch.items()) + list(write.items()):
    if "tilespmv" in k:
        per.setdefault(k, {})[c + "_kb"] = v
total = 0
for k, t in per.items():
    t["bytes_per_launch"] = int((t.get("FETCH_SIZE_kb", 0) * factor + t.get("WRITE_SIZE_kb", 0)) * 1024)
# a SpMV of a panelled / dense-tile plan is several launches: bytes per SpMV = sum over the kernels of bytes per launch x launches per SpMV (k_units runs once per SpMV)
calls = {k["name"]: k["calls"] for k in kern}
unit_calls = max([c for n, c in calls.items() if "k_units<" in n or "k_tiles_direct" in n] or [0])
per_spmv = None
if unit_calls:
    per_spmv = 0
    for k, t in per.items():
        c = calls.get(k) or calls.get("void " + k) or 0
        per_spmv += int(t["bytes_per_launch"] * c / unit_calls)
bench = None
for line in open(os.path.join(d, "bench_under_trace.json")):
    if line.startswith("{"):
        bench = json.loads(line)
if bench is not None and isinstance(bench.get("roofline"), dict):
    # the line printed under the trace looked at the PREVIOUS traffic file (this one did not exist yet): what it says about that file's freshness is not about this one
    for k in ("traffic", "traffic_source", "actual_traffic_gbps", "actual_traffic_over_read_ceiling"):
        bench["roofline"].pop(k, None)
out = {"workload": wl, "dtype": dt, "kernel": dom, "measured": time.strftime(os.environ.get("TILESPMV_ROUND_TAG", "round 4") + ", %Y-%m-%d"),
       # the plan these passes measured: bench.py reports `traffic` only while its live plan has the same fingerprint
       "plan_fingerprint": None if bench is None else (bench.get("roofline") or {}).get("plan_fingerprint"),
       "FETCH_SIZE_kb": per.get(dom, {}).get("FETCH_SIZE_kb"), "WRITE_SIZE_kb": per.get(dom, {}).get("WRITE_SIZE_kb"),
       "fetch_correction": factor, "hbm_bytes_per_launch": per.get(dom, {}).get("bytes_per_launch"), "hbm_bytes_per_spmv": per_spmv,
       "launches_per_spmv": {k: round(c / unit_calls, 3) for k, c in calls.items()} if unit_calls else None,
       "all_kernels_bytes_per_launch": {k: t["bytes_per_launch"] for k, t in per.items()},
       "kernel_stats": kern,
       "bench_under_trace": None if bench is None else {k: bench.get(k) for k in ("value", "ms_per_step", "roofline")},
       "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --workload %s --dtype %s` (scripts/profile_traffic.sh); FETCH_SIZE x the factor measured on a known 1-GiB read in the same access pattern; requests at the L2<->fabric boundary, Infinity-Cache hits included" % (wl, dt)}
json.dump(out, open(os.path.join(d, "traffic_%s_%s.json" % (wl, dt)), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("workload", "dtype", "kernel", "hbm_bytes_per_launch", "fetch_correction")}))
for k in kern[:4]:
    print("   %-50s calls %5d avg %10.1f ns" % (k["name"][-50:], k["calls"], k["avg_ns"]))
