#!/usr/bin/env python3
"""Print per-launch durations of half_sweep_kernel from a rocprofv3 kernel_trace.csv (development aid)."""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "half_sweep" in r["Kernel_Name"]]
for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"{dur:10.3f} ms  grid={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):7d} wg={r['Workgroup_Size_X']:>4s} lds={r['LDS_Block_Size']:>7s} vgpr={r['VGPR_Count']:>4s} {r['Kernel_Name'][:60]}")
