#!/bin/bash
# usage (GPU box): scripts/exp_variants.sh <tag> "<bench args>" <variant dir | .> ...   -- the same bench (--no-cpu --no-extra) on this tree (.) and on
# variant builds under _ab/ (scripts/build_variant.sh; their POISMF_HIP_EXTRA_FLAGS must be given again as FLAGS_<name> in the environment)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
ARGS=$1; shift
for v in "$@"; do
  d=$R; name=main; fl=""
  if [ "$v" != "." ]; then d=$R/_ab/$v; name=$v; eval fl=\$FLAGS_$v; fi
  POISMF_HIP_EXTRA_FLAGS="$fl" python3 $d/bench.py --no-cpu --no-extra $ARGS > $OUT/$name.log 2>&1
  grep '^{"metric"' $OUT/$name.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('$name', 'ms', round(d['ms_per_step'],3), 'B', round(r['kernel_ms_B_half'],3), 'A', round(r['kernel_ms_A_half'],3), 'frac', round(r['frac'],3))
    for L in r['launches']: print('   ', L['half'], L['kernel'], L['rows'], round(L['avg_ms'],3), round(L['frac'],3))
"
done
