#!/usr/bin/env python3
"""mean per dispatch of every counter in the rocprofv3 --pmc csv files under a directory, per kernel"""
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_nrt" not in k and len(sys.argv) < 3: continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k[:90])
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
