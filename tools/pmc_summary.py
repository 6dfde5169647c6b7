#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs written by tools/pmc_passes.sh: per kernel, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        if filt and filt not in row["Kernel_Name"]:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} mean/dispatch {sum(v)/len(v):16.1f}   n={len(v)}")
