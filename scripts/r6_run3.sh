#!/bin/bash
# round 6, third GPU batch: the GPU suite on the pruned tree, then the round's profiles (scripts/profile_round.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/tests_gpu.log
tail -3 $O/tests_gpu.log
scripts/profile_round.sh r6h r06 > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log
cat $O/bench_line.json | cut -c1-1500
