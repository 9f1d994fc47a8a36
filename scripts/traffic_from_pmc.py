#!/usr/bin/env python3
"""traffic_from_pmc.py <dir> <sweeps>: fabric-side bytes per full sweep of the row-update kernels, from the
summary.txt files scripts/profile_round.sh leaves in <dir>/pmc_f*/ (FETCH_SIZE) and <dir>/pmc_w*/ (WRITE_SIZE).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB, and on gfx950 FETCH_SIZE tallies the 128-byte
requests of 16-byte-per-lane loads at 64 bytes (MI355X_MICROARCH.md, section HBM).  Infinity-Cache hits are included,
so this is an upper bound on DRAM traffic."""
import glob
import json
import os
import re
import sys

root, sweeps = sys.argv[1], float(sys.argv[2])


def total(path, counter):
    tot, name = 0.0, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        elif name and "half_sweep" in name and counter in line:
            tot += float(re.search(r"sum=([0-9.e+]+)", line).group(1))
    return tot


out = {"_note": "fabric-side bytes per full sweep of the half_sweep_* launches on config C2, (2 * FETCH_SIZE + WRITE_SIZE) * 1024, "
                "FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes (scripts/profile_round.sh); Infinity-Cache hits "
                "are included, so this is an upper bound on DRAM traffic"}
for tag, key in (("10", "pg_maxupd10_f32"), ("1", "pg_maxupd1_f32")):
    f = os.path.join(root, f"pmc_f{tag}", "**", "summary.txt")
    w = os.path.join(root, f"pmc_w{tag}", "**", "summary.txt")
    ff, ww = glob.glob(f, recursive=True), glob.glob(w, recursive=True)
    if ff and ww:
        out[key] = (2 * total(ff[0], "FETCH_SIZE") + total(ww[0], "WRITE_SIZE")) * 1024 / sweeps
print(json.dumps(out, indent=1))
