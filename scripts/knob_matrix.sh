#!/bin/bash
# usage (on the GPU box): scripts/knob_matrix.sh -- the GPU parity suites once under every testing / tuning knob, so that the
# code paths the defaults no longer take (LDS engine for short rows, compact gathers, generic slot counts, ...) stay green
for knob in POISMF_HIP_NO_PAD POISMF_HIP_NO_PREFETCH POISMF_HIP_GENERIC POISMF_HIP_NO_REGTILE POISMF_HIP_STATIC_ROWS \
            POISMF_HIP_NO_FORK POISMF_HIP_FORK_BINS POISMF_HIP_CG_CACHE_RESIDENT POISMF_HIP_CG_NOCACHE POISMF_HIP_NO_LONGROW POISMF_HIP_NO_STREAM_CACHE POISMF_HIP_NO_TEAM \
            POISMF_HIP_NO_LANE POISMF_HIP_NO_LS_PRUNE POISMF_HIP_NO_STAGED_UPLOAD POISMF_HIP_PG_LONG_LANE POISMF_HIP_K100_LANE_B POISMF_HIP_NO_ARRIVE_WAIT \
            POISMF_HIP_ONE_DMA_QUEUE POISMF_HIP_NO_UPLOAD_OVERLAP POISMF_HIP_PG_LANE_ROWS=1 POISMF_HIP_PG_LANE_ROWS=2 POISMF_HIP_DEVICE_CACHE_MB=4096 POISMF_HIP_GRID_MULT=1 \
            POISMF_HIP_K50_P32=0 POISMF_HIP_NO_TX POISMF_HIP_TX_MAX=48 POISMF_HIP_K100_MID=1 POISMF_HIP_K100_LANE_MAX=128 POISMF_HIP_K100_LANE_B=0 POISMF_HIP_NO_GIANT_TEAMS POISMF_HIP_NO_LANE_TEAMS POISMF_HIP_LANE_TEAM_STREAM=0 POISMF_HIP_NO_ROW_INTERRUPT; do
  case $knob in *=*) spec=$knob ;; *) spec=$knob=1 ;; esac
  echo "== $spec"
  env $spec timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_regtile.py tests/test_gpu_rows.py tests/test_gpu_team.py tests/test_gpu_giant.py tests/test_gpu_decisions.py -m gpu -q -x 2>&1 | tail -1
done
