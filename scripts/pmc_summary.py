#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch
(only dispatches longer than --min-grid workgroups, to look at the main row-bin launches)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(path)):
    if int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0) < min_grid:
        continue
    name = r["Kernel_Name"].split("(")[0][-110:]   # (long enough for half_sweep_lane_kernel<...> with its ten arguments)
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):.6g} sum={sum(v):.6g}")
