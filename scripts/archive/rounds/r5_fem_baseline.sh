#!/bin/bash
# Round 5: the FEM / block-structured class on the engine as round 4 left it (default plan, whole CSR tiles) — the "before" of the pooled-unit form.
set -e
mkdir -p gpurun_out/r5base
for wl in fem3_68 fem6_46 fem3s64_68; do
  for split in default 0; do
    if [ "$split" = default ]; then unset TILESPMV_CSR_SPLIT; else export TILESPMV_CSR_SPLIT=$split; fi
    timeout -k 10 300 python bench.py --workload $wl --steps 50 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/r5base/${wl}_split${split}.json 2> gpurun_out/r5base/${wl}_split${split}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r5base/${wl}_split${split}.json"))
r=d["roofline"]; print("${wl} csr_split=${split}: %.4f ms frac %.3f min %.3f plan %.3f entry_mode %s check %s"%(d["ms_per_step"], r["frac"], r["frac_min_bytes"], r["frac_by_plan_bytes"], d["config"]["entry_mode"], d["check"][:4]))
PY
  done
done
