#!/bin/bash
# usage (GPU box): scripts/exp_pair.sh <tag>  -- PG fp32 headline with two row streams per workgroup (default) and with one (POISMF_HIP_NO_PAIR=1), alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
run() {
  name=$1; shift
  env "$@" python3 $R/bench.py --no-cpu --no-extra --steps 10 --warmup 3 > $OUT/pg_$name.log 2>&1
  grep '^{"metric"' $OUT/pg_$name.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('$name', 'ms', round(d['ms_per_step'],3), 'B', round(r['kernel_ms_B_half'],3), 'A', round(r['kernel_ms_A_half'],3), 'frac', round(r['frac'],3))
    for L in r['launches']: print('   ', L['half'], L['kernel'], L['rows'], round(L['avg_ms'],3), round(L['frac'],3))
"
}
run pair POISMF_X=0
run single POISMF_HIP_NO_PAIR=1
run pair2 POISMF_X=0
run single2 POISMF_HIP_NO_PAIR=1
