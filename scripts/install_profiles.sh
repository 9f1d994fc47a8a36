#!/bin/bash
# usage: scripts/install_profiles.sh <tag> [round dir, default r01]: copy what scripts/profile_round.sh left in
# gpurun_out/<tag>/ into profiles/<round>/ (tracked) and refresh profiles/hbm_traffic.json
set -e
cd "$(dirname "$0")/.."
SRC=gpurun_out/$1; DST=profiles/${2:-r01}
rm -rf $DST; mkdir -p $DST/pmc
for d in kt_default kt_pg10 kt_pg1 kt_cg64; do cp $SRC/$d/kt_kernel_stats.csv $DST/${d}_kernel_stats.csv; done
cp $SRC/bench_line.json $DST/bench_line.json
cp $SRC/bench_line_under_rocprof.json $DST/kt_default_bench_line.json
for d in kt_pg10 kt_pg1 kt_cg64; do cp $SRC/${d}_bench_line.json $DST/; done
for d in pmc_f10 pmc_w10 pmc_t10 pmc_sq10 pmc_f1 pmc_w1; do cp $SRC/$d/summary.txt $DST/pmc/${d}.summary.txt; done
cp $SRC/hbm_traffic.json profiles/hbm_traffic.json
python3 - <<PY
import csv, json
for tag, n in (("kt_pg10", 6), ("kt_pg1", 6), ("kt_cg64", 4)):
    rows = list(csv.DictReader(open(f"$DST/{tag}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows if "half_sweep" in r["Name"])
    d = json.loads(open(f"$DST/{tag}_bench_line.json").read())
    print(f"{tag}: rocprof sum(half_sweep_*) / sweeps = {tot / 1e6 / n:.4f} ms   bench kernel_ms_per_sweep = {d['roofline']['kernel_ms_per_sweep']:.4f} ms")
PY
