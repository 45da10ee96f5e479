#!/bin/bash
# round 4: does the cache policy of the scattered x gathers change how many bytes a miss pulls across the fabric?  (diagnostic builds _g1 nt, _g2 sc1, _g3 sc0 sc1)
set -o pipefail
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r4gather; mkdir -p $out
for wl in uniform8_4000000 bandrand4x3_2000000 powerlaw8000000; do
  timeout -k 10 400 python scripts/exp_bench.py $wl Q=0 LIB=_g1 LIB=_g2 LIB=_g3 > $out/exp_$wl.txt 2>&1; echo "== $wl rc=$?"; grep -v "amdgpu.ids" $out/exp_$wl.txt | cut -c1-200
done
cd /tmp
for v in "" _g1 _g2; do
  for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    TILESPMV_LIB_VARIANT=$v timeout -k 5 240 rocprofv3 --pmc $c --output-format csv -d $out/pmc${v}_$tag -- python $GRAFT_REPO_ROOT/bench.py --workload uniform8_4000000 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-check > $out/pmc${v}_$tag.log 2>&1 || echo "pmc $v $c failed"
  done
  python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out/pmc${v}_ > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4gather"
for d in sorted(glob.glob(out + "/pmc*_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_units" in row.get("Kernel_Name", ""): acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
