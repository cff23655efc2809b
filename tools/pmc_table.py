#!/usr/bin/env python3
"""Print per-kernel means of every counter in rocprofv3 --pmc output dirs. usage: pmc_table.py DIR [DIR...] [--kernel SUBSTR]"""
import collections, csv, glob, os, sys
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
filt = None
if "--kernel" in sys.argv: filt = sys.argv[sys.argv.index("--kernel") + 1]; dirs = [d for d in dirs if d != filt]
agg = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if filt and filt not in k: continue
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, cn), v in sorted(agg.items()):
    print(f"{k:32s} {cn:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
