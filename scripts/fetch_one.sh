#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per k_units launch for one bench workload: scripts/fetch_one.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/fetch1_$tag
mkdir -p $out; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-check "$@" > $out/$c.log 2>&1 || echo "$c failed"
done
python - "$out" <<'PY'
import sys, glob, csv, collections, json
d = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tilespmv" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    mean = sum(v) / len(v)
    print("  %-42s %-10s launches %d  mean %.0f KB -> %.1f MB" % (k, c, len(v), mean, mean * (1.99996 if c == "FETCH_SIZE" else 1.0) * 1024 / 1e6))
for line in open(d + "/FETCH_SIZE.log"):
    if line.startswith("{"):
        j = json.loads(line); print("  bench: ms %.4f  B_alg %.1f MB  plan stream bytes %.1f MB" % (j["ms_per_step"], j["roofline"]["algorithmic_bytes_per_launch"] / 1e6, j["roofline"]["plan_stream_bytes_per_launch"] / 1e6))
PY
