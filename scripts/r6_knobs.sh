#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6k; mkdir -p $O; cd $R
scripts/knob_matrix.sh > $O/knob_matrix.txt 2>&1
cat $O/knob_matrix.txt
