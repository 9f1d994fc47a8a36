#!/bin/bash
# usage (GPU box): scripts/exp_cg64.sh <tag>   -- CG fp64 on the C4/C3 matrix under a few knobs, one JSON line each
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu --no-extra --method cg --fp64 --steps 2 --warmup 1"
$B > $OUT/cg64_default.log 2>&1
POISMF_HIP_LONGROW_NNZ=512 $B > $OUT/cg64_long512.log 2>&1
for f in $OUT/cg64_*.log; do echo "== $f"; grep '^{"metric"' $f | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print(d['ms_per_step'], r['kernel_ms_B_half'], r['kernel_ms_A_half'], r['frac'], r['pass_weighted']['tile_passes_per_row'], r['kernel'][:400])
"; done
