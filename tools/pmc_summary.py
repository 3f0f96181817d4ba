#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter per dispatch."""
import collections, csv, glob, os, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0][:48]
        agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in sorted(agg.items()):
    if name.startswith(("void at::", "__amd")):
        continue
    print(name)
    for c, vals in sorted(ctrs.items()):
        vals = sorted(vals)
        print("   %-24s n=%-4d max=%.4g median=%.4g sum=%.6g" % (c, len(vals), vals[-1], vals[len(vals) // 2], sum(vals)))
