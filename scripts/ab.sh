#!/bin/bash
# usage (GPU box): scripts/ab.sh <args for bench.py>   -- the same bench on the tree under _ab/old and on this tree, alternating, same box
# (_ab/old: `git worktree add _ab/old <commit>` and `python -m poismf_amd.build` inside it, here, before the gpurun call; git-ignored)
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for t in _ab/old .; do
    python3 $R/$t/bench.py --no-cpu --no-extra "$@" 2>/dev/null | grep '^{"metric"' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('$t', 'ms', round(d['ms_per_step'],3), 'B', round(r['kernel_ms_B_half'],3), 'A', round(r['kernel_ms_A_half'],3), 'frac', round(r['frac'],3))
"
  done
done
