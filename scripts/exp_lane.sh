#!/bin/bash
# usage (GPU box): scripts/exp_lane.sh <tag>  -- CG fp64 on the C3 matrix: maxupd 0/1/2/5 (fixed cost per row vs cost per iteration),
# lane engine on / off, then SQ counters of the default run
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu --no-extra --method cg --fp64 --steps 2 --warmup 1"
for mu in 1 2 5; do $B --maxupd $mu > $OUT/cg64_mu$mu.log 2>&1; done
POISMF_HIP_NO_LANE=1 $B > $OUT/cg64_nolane.log 2>&1
for f in $OUT/cg64_*.log; do echo "== $f"; grep '^{"metric"' $f | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['ms_per_step'], r['kernel_ms_B_half'], r['kernel_ms_A_half'], r['frac'], r['pass_weighted']['tile_passes_per_row'])
    for L in r['launches']: print('    ', L['kernel'], L['half'], L['rows'], round(L['avg_ms'], 3), round(L['frac'], 3))
"; done
bash $R/scripts/pmc_run.sh $1/pmc --method cg --fp64 --steps 2 --warmup 1 > $OUT/pmc.log 2>&1
cat $OUT/pmc/sq/*/*.summary.txt $OUT/pmc/sq2/*/*.summary.txt 2>/dev/null | grep -A9 "lane_kernel" | head -60
