#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the unit kernel for the product library and diagnostic variants: scripts/fetch_ab.sh <workload> <dtype> "<variant> <variant> ..."  ("-" = product)
wl=$1; dt=$2; export TMPDIR=/tmp
for v in $3; do
  [ "$v" = "-" ] && v=""
  export TILESPMV_LIB_VARIANT=$v
  out=$GRAFT_REPO_ROOT/gpurun_out/fetch_ab_${wl}${v}; mkdir -p $out
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc_$c
    (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-check --workload $wl --dtype $dt --no-cpu-baseline --no-extras > /dev/null 2> $out/$c.err) || echo "$c failed"
    python3 - $out/pmc_$c $c "$v" <<'P'
import sys, glob, csv
d, c, v = sys.argv[1:4]
f = glob.glob(d + "/*/*counter_collection.csv")[0]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_units<" in r["Kernel_Name"] and r["Counter_Name"] == c]
print("variant '%s' %s per k_units launch: %.1f MB (%d launches)" % (v, c, sum(vals) / len(vals) * (2.0 if c == "FETCH_SIZE" else 1.0) * 1024 / 1e6, len(vals)))
P
  done
done
