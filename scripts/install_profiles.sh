#!/bin/bash
# usage: scripts/install_profiles.sh <tag> [round dir, default r06]: copy what scripts/profile_round.sh left in
# gpurun_out/<tag>/ into profiles/<round>/ (tracked) and refresh profiles/hbm_traffic.json.  Only the files this script writes are
# replaced: whatever else lives in the round's directory (probe outputs, run_config records, kernel_resource_usage.txt) stays.
set -e
cd "$(dirname "$0")/.."
SRC=gpurun_out/$1; DST=profiles/${2:-r06}
mkdir -p $DST/pmc
cp $SRC/bench_line.json $DST/bench_line.json
[ -f $SRC/bench_full.json ] && cp $SRC/bench_full.json $DST/bench_full.json
for n in pg10 pg1 cg64 cg32 tncg32 c5; do
  [ -f $SRC/kt_$n/kt_kernel_stats.csv ] || continue
  cp $SRC/kt_$n/kt_kernel_stats.csv $DST/kt_${n}_kernel_stats.csv
  [ -f $SRC/kt_${n}_bench_line.json ] && cp $SRC/kt_${n}_bench_line.json $DST/
  [ -f $SRC/kt_${n}_run_config.json ] && cp $SRC/kt_${n}_run_config.json $DST/
  for c in f w t sq; do f=$(find $SRC/pmc_${c}_$n -name summary.txt 2>/dev/null | head -1); [ -n "$f" ] && cp $f $DST/pmc/pmc_${c}_${n}.summary.txt; done
done
[ -f $SRC/kt_c5_timeline.txt ] && cp $SRC/kt_c5_timeline.txt $DST/kt_c5_timeline.txt
cp $SRC/hbm_traffic.json profiles/hbm_traffic.json
python3 - <<PY
import csv, json, os
for tag, n in (("kt_pg10", 12), ("kt_pg1", 12), ("kt_cg64", 6), ("kt_cg32", 6), ("kt_tncg32", 6)):
    if not os.path.exists(f"$DST/{tag}_kernel_stats.csv"):
        continue
    rows = list(csv.DictReader(open(f"$DST/{tag}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows if "half_sweep" in r["Name"])
    d = json.loads(open(f"$DST/{tag}_bench_line.json").read())
    print(f"{tag}: rocprof sum(half_sweep_*) / sweeps = {tot / 1e6 / n:.4f} ms   bench kernel_ms_per_sweep = {d['roofline']['kernel_ms_per_sweep']:.4f} ms")
PY
