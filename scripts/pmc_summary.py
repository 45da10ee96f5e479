"""Summarise rocprofv3 --pmc CSVs: mean counter value per dispatch of each tilespmv kernel."""
import csv, glob, os, sys, json
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "tilespmv" not in name: continue
        short = name.split("(")[0].replace("void tilespmv::", "")
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: (sum(v) / len(v)) for c, v in sorted(cs.items())}
    out[k]["_dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
