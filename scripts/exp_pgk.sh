#!/bin/bash
# usage (GPU box): scripts/exp_pgk.sh  -- PG fp32 on the 1e8-nnz matrix for a few inner-update counts: fixed cost per row vs cost per pass
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in 0 1 2 4 10; do
  python3 $R/bench.py --no-cpu --no-extra --steps 5 --warmup 2 --maxupd $m 2>/dev/null | grep '^{"metric"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('maxupd $m', 'ms', round(d['ms_per_step'],3), 'B', round(r['kernel_ms_B_half'],3), 'A', round(r['kernel_ms_A_half'],3))
"
done
