#!/usr/bin/env python3
"""Development aid: start / end (ms, relative to the first) of every half_sweep_* launch of a rocprofv3 --kernel-trace CSV: which
launches of a half-sweep actually ran beside which.  usage: trace_timeline.py <rocprof output dir> [last N launches]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = [r for r in csv.DictReader(open(f)) if any(t in r["Kernel_Name"] for t in ("half_sweep", "team_", "hold_back", "colsum"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print("%9.3f .. %9.3f ms (%8.3f)  queue=%s grid=%s lds=%s  %s" % (a, b, b - a, r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:90]))
