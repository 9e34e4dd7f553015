#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import collections
import csv
import sys

for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(path.split("gpurun_out/")[-1], "|", k)
        for c, v in cs.items():
            print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
