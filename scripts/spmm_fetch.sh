#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the multi-vector kernels: scripts/spmm_fetch.sh <tag> <workload> <dtype> <nvec list> (env passes through)
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/spmmfetch_$tag; mkdir -p $out; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/$c -- python $GRAFT_REPO_ROOT/scripts/spmm_bench.py "$@" > $out/$c.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, collections, sys, json
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tilespmv" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-30:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    m = sum(v) / len(v)
    print("   %-32s %-10s %.1f MB" % (k, c, m * (2.0 if c == "FETCH_SIZE" else 1.0) * 1024 / 1e6))
for l in open(sys.argv[1] + "/FETCH_SIZE.log"):
    if l.startswith("{"):
        print("   (under pmc)", [(r["nvec"], r["ms"]) for r in json.loads(l)["results"]])
PY
