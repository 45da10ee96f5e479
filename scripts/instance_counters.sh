#!/bin/bash
# per-instance address-translation counters of N identical plans (scripts/instance_counters.py): one --pmc pass per counter pair + a kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/instctr; rm -rf $out; mkdir -p $out
wl=${1:-nlpkkt160}; dt=${2:-f64}
timeout -k 10 300 python scripts/instance_counters.py $wl $dt 4 10 2>&1 | grep -v amdgpu.ids
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python $GRAFT_REPO_ROOT/scripts/instance_counters.py $wl $dt 4 10 > $out/trace.log 2>&1
SETS=${SETS:-"TCP_UTCL1_TRANSLATION_MISS_sum+TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY+GRBM_GUI_ACTIVE"}
for set in $SETS; do set=$(echo $set | tr '+' ' ')
  tag=$(echo $set | tr ' ' '+')
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $out/$tag -- python $GRAFT_REPO_ROOT/scripts/instance_counters.py $wl $dt 4 10 > $out/$tag.log 2>&1 || echo "$tag failed"
done
python - $out <<'PY'
import sys, glob, csv, collections
d = sys.argv[1]
# kernel trace: durations of the last 40 k_units dispatches, in order
rows = []
for f in glob.glob(d + "/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_units" in r["Kernel_Name"]: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort(); last = rows[-40:]
print("kernel trace, ns per SpMV by instance:", [round(sum(x[1] for x in last[i*10:(i+1)*10]) / 10) for i in range(4)])
for sub in sorted(glob.glob(d + "/*+*")):
    if not sub.endswith(".log"):
        acc = collections.defaultdict(list)
        for f in glob.glob(sub + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "k_units" in r["Kernel_Name"]: acc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in acc.items():
            v.sort(); last = v[-40:]
            print("%-44s by instance: %s" % (c, [round(sum(x[1] for x in last[i*10:(i+1)*10]) / 10) for i in range(4)]))
PY
