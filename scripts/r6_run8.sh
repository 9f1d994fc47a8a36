#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6j; mkdir -p $O; cd $R
scripts/profile_round.sh r6j r06 > $O/profile_round.log 2>&1
tail -3 $O/profile_round.log
