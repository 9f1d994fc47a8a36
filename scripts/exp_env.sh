#!/bin/bash
# usage (GPU box): scripts/exp_env.sh <tag> <bench args or ""> NAME=ENV=VAL[,ENV=VAL..] ...   -- the bench (--no-cpu --no-extra) once per named environment
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
ARGS=$1; shift
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  env $(echo $envs | tr ',' ' ') python3 $R/bench.py --no-cpu --no-extra --steps 10 --warmup 3 $ARGS > $OUT/$name.log 2>&1
  grep '^{"metric"' $OUT/$name.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('$name', 'ms', round(d['ms_per_step'],3), 'B', round(r['kernel_ms_B_half'],3), 'A', round(r['kernel_ms_A_half'],3), 'frac', round(r['frac'],3), 'valu', round(r.get('valu',{}).get('frac',0),3))
    for L in r['launches']: print('   ', L['half'], L['kernel'], L['rows'], round(L['avg_ms'],3), round(L['frac'],3))
"
done
