"""Prints the preparation records of a bench line (argv[1]): headline prep_seconds and, per other workload, host Tile_create / plan create against the device pipeline."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline:", d["value"], d["unit"], d["ms_per_step"], "ms/step, frac", d["roofline"]["frac"])
print("prep_seconds:", json.dumps(d["prep_seconds"]))
ow = d.get("other_workloads", {})
for k, v in (ow.items() if isinstance(ow, dict) else []):
    if not isinstance(v, dict): continue
    plan = v.get("default_plan") or v.get("coo_in_tile") or {}
    print("%-28s host Tile_create %s s + plan %s s | on device %s | %s ms frac %s" % (k, v.get("tile_create_seconds"), plan.get("plan_create_seconds"), json.dumps(v.get("prepared_on_device")), plan.get("ms_per_spmv"), plan.get("frac_of_8TBps")))
