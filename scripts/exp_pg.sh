#!/bin/bash
# usage (GPU box): scripts/exp_pg.sh <tag> [env assignments...]  -- PG fp32 headline under a few run-time knobs, one line each
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
run() {
  name=$1; shift
  env "$@" python3 $R/bench.py --no-cpu --no-extra --steps 5 --warmup 2 > $OUT/pg_$name.log 2>&1
  grep '^{"metric"' $OUT/pg_$name.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('$name', 'ms', round(d['ms_per_step'],3), 'B', round(r['kernel_ms_B_half'],3), 'A', round(r['kernel_ms_A_half'],3), 'frac', round(r['frac'],3))
"
}
run default POISMF_X=0
run gm1 POISMF_HIP_GRID_MULT=1
run gm4 POISMF_HIP_GRID_MULT=4
run gm8 POISMF_HIP_GRID_MULT=8
run gm64 POISMF_HIP_GRID_MULT=64
