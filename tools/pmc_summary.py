#!/usr/bin/env python3
"""Per kernel (and grid size) averages of every counter found in rocprofv3 --pmc counter_collection CSVs.
usage: pmc_summary.py <dir or csv> [<dir or csv> ...] [--match substr]"""
import csv, glob, json, os, sys
from collections import defaultdict
args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else "icpmi"
args = [a for a in args if a != match]
files = []
for a in args:
    files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]:
            continue
        wg = int(r["Workgroup_Size"])
        key = f'{r["Kernel_Name"].split("(")[0][:70]} [{int(r["Grid_Size"]) // max(wg, 1)} wg x {wg}]'
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    v = d.get("SQ_INSTS_VALU"); 
    if v and d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_THREAD_CYCLES_VALU"):
        d["lanes_per_valu"] = round(d["SQ_THREAD_CYCLES_VALU"] / d["SQ_ACTIVE_INST_VALU"], 1)
print(json.dumps(out, indent=1))
