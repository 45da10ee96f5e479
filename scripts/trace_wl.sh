#!/bin/bash
# per-kernel durations for one workload/variant: scripts/trace_wl.sh <workload> <variant...>
export TMPDIR=/tmp
wl=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$wl
rm -rf $out; mkdir -p $out; cd /tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/scripts/exp_bench.py $wl "$@" > $out/log.txt 2>&1
grep -v amdgpu.ids $out/log.txt | tail -5
python - <<PY
import csv, glob
for f in glob.glob("$out/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "tilespmv" in r["Name"]:
            print("%-60s calls %6s avg %9.1f ns  total %5.1f%%" % (r["Name"].split("(")[0].replace("void tilespmv::",""), r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
