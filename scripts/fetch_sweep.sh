#!/bin/bash
# FETCH_SIZE per k_units launch under different workgroup->XCD mappings (one rocprofv3 --pmc pass each).
# usage: scripts/fetch_sweep.sh <tag> "ENV=VAL ..." "ENV=VAL ..." ...
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/fetch_$tag
mkdir -p $out
cd /tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "cfg$i: $cfg"
  ( export $cfg; timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cfg$i -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-check > $out/cfg$i.log 2>&1 ) || echo "cfg$i failed"
  python - "$out/cfg$i" "$cfg" <<'PY'
import sys, glob, csv, collections
d, cfg = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_units" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("  %-40s launches %d  FETCH_SIZE mean %.0f KB -> x1.99996 = %.1f MB per launch" % (k, len(v), sum(v) / len(v), sum(v) / len(v) * 1.99996 * 1024 / 1e6))
PY
done
