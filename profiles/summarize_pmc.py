#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv + kernel_trace.csv into a per-kernel table.
usage: summarize_pmc.py <dir-with-*_counter_collection.csv> [...]"""
import csv, glob, sys, collections, json
def short(n):
    n = n.replace("void sicn::", "").replace("sicn::", "")
    return n.split("(")[0]
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            name = short(r["Kernel_Name"])
            if "k_" not in name: continue
            key = (name, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for key, cs in acc.items():
            o = out.setdefault(" ".join(key), {})
            for c, v in cs.items():
                o[c] = sum(v) / len(v)
                o["_n"] = len(v)
print(json.dumps(out, indent=1))
