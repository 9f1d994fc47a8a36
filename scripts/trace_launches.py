#!/usr/bin/env python3
"""Development aid: print every half_sweep_* launch of a rocprofv3 --kernel-trace CSV with its duration and geometry."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "half_sweep" in r["Kernel_Name"]]
keys = rows[0].keys() if rows else []
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    geo = " ".join(f"{k}={r[k]}" for k in keys if k in ("Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "Scratch_Size"))
    print("%9.3f ms  %s  %s" % (d, geo, r["Kernel_Name"][:60]))
