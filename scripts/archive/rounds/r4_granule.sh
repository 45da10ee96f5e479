#!/bin/bash
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r4granule; mkdir -p $out; cd /tmp
$GRAFT_REPO_ROOT/scripts/micro/gather_granule > $out/timing.txt 2>&1; cat $out/timing.txt
for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 5 120 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$tag -- $GRAFT_REPO_ROOT/scripts/micro/gather_granule > $out/pmc_$tag.log 2>&1 || echo "pmc $c failed"
done
python - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4granule"
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k in sorted(acc): print(k, [round(v) for v in acc[k]][:8])
PY
