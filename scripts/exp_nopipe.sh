#!/bin/bash
# usage (GPU box): scripts/exp_nopipe.sh -- the row hand-out modes of PG's multi-wave lane launch (POISMF_HIP_PG_LANE_ROWS) and CG fp32, on this
# tree and on the variant without sweep_rows' cross-row pipeline for multi-wave lane rows (scripts/build_variant.sh nopipe f -DPMF_LANE_NO_PIPE=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
export FLAGS_nopipe="-DPMF_LANE_NO_PIPE=1"
for rows in 0 1 2; do
  echo "== PG(10), POISMF_HIP_PG_LANE_ROWS=$rows"
  POISMF_HIP_PG_LANE_ROWS=$rows $R/scripts/exp_variants.sh np_pg$rows "--steps 10 --warmup 3" . nopipe 2>&1 | grep " ms \|lane_kernel"
done
echo "== CG fp32"
$R/scripts/exp_variants.sh np_cg "--steps 3 --warmup 1 --method cg" . nopipe 2>&1 | grep " ms \|lane_kernel.*NW=8"
