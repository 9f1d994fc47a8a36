#!/usr/bin/env python3
"""traffic_from_pmc.py <dir>: fabric-side bytes per full sweep of the row-update kernels, from the summary.txt files
scripts/profile_round.sh leaves in <dir>/pmc_f_*/ (FETCH_SIZE) and <dir>/pmc_w_*/ (WRITE_SIZE).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB, and on gfx950 FETCH_SIZE tallies the 128-byte
requests of 16-byte-per-lane loads at 64 bytes (MI355X_MICROARCH.md, section HBM).  Infinity-Cache hits are included,
so this is an upper bound on DRAM traffic."""
import glob
import json
import os
import re
import sys

root = sys.argv[1]
round_tag = sys.argv[2] if len(sys.argv) > 2 else "?"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poismf_amd import build  # noqa: E402  (the hash of the kernel sources + flags these counters were recorded from)
# name -> (key in profiles/hbm_traffic.json, sweeps per profiled run = 2 x (warmup + steps) of scripts/profile_round.sh)
RUNS = {"pg10": ("C4_pg_maxupd10_f32", 12), "pg1": ("C4_pg_maxupd1_f32", 12), "cg64": ("C4_cg_maxupd5_f64", 6),
        "cg32": ("C4_cg_maxupd5_f32", 6), "tncg32": ("C4_tncg_maxupd750_f32", 6),
        # config C5 through scripts/run_config.py (--warmup 2 --sweeps 3): five sweeps, the first two from the random start
        "c5": ("C5_tncg_maxupd1500_f64", 5)}


def total(path, counter):
    tot, name = 0.0, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        elif name and ("half_sweep" in name or "eep_lane_kernel" in name) and counter in line:   # (summaries cut names to their tail)
            tot += float(re.search(r"sum=([0-9.e+]+)", line).group(1))
    return tot


out = {"_note": "fabric-side bytes per full sweep of the half_sweep_* launches on the 1M x 100K / 1e8-nnz matrix (C5_*: on config C5's "
                "matrix, averaged over the five sweeps of the profiled run), (2 * FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE and WRITE_SIZE "
                "from separate rocprofv3 --pmc passes (scripts/profile_round.sh); Infinity-Cache hits are included, so this is an upper "
                "bound on DRAM traffic.  source_hash = poismf_amd.build's hash of the kernel sources + flags the counters were recorded "
                "from: bench.py quotes an entry as roofline.traffic only while the tree still has that hash",
       "round": round_tag, "source_hash": build._source_hash()}
for name, (key, sweeps) in RUNS.items():
    # the key bench.py looks up: <workload>_<method>_maxupd<N>_<f32|f64>, read off the bench line of the same run when it is there
    try:
        d = json.loads(open(os.path.join(root, f"kt_{name}_bench_line.json")).read())
        m = re.search(r"method=(\w+), maxupd=(\d+)", d["config"]["workload"])
        key = f"C4_{m.group(1)}_maxupd{m.group(2)}_{d['dtype']}"
    except Exception:
        pass
    ff = glob.glob(os.path.join(root, f"pmc_f_{name}", "**", "summary.txt"), recursive=True)
    ww = glob.glob(os.path.join(root, f"pmc_w_{name}", "**", "summary.txt"), recursive=True)
    if ff and ww:
        out[key] = (2 * total(ff[0], "FETCH_SIZE") + total(ww[0], "WRITE_SIZE")) * 1024 / sweeps
print(json.dumps(out, indent=1))
