#!/bin/bash
# usage (GPU box): scripts/exp_f32lane.sh <tag> -- fp32 CG / TNCG with and without the lane engine (C2), CG fp32 on the C4 matrix, C5 TNCG fp64
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
for lane in on off; do
  E=""; [ $lane = off ] && E="POISMF_HIP_NO_LANE=1"
  env $E python3 $R/scripts/run_config.py C2 --method cg --fp32 --sweeps 3 > $OUT/c2_cg32_$lane.log 2>&1
  env $E python3 $R/scripts/run_config.py C2 --method tncg --fp32 --sweeps 3 > $OUT/c2_tncg32_$lane.log 2>&1
  env $E python3 $R/scripts/run_config.py C4 --method cg --fp32 --sweeps 3 --sample 0 > $OUT/c4_cg32_$lane.log 2>&1
  env $E python3 $R/scripts/run_config.py C5 --sweeps 3 --sample 100 > $OUT/c5_$lane.log 2>&1
done
for f in $OUT/*.log; do echo "== $f"; tail -3 $f | cut -c1-600; done
