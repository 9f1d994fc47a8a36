#!/bin/bash
# usage (GPU box): scripts/knob_round5.sh -- the GPU suites that reach the round-5 paths, once under every knob that switches one of them off or over
# (the full matrix of every knob is scripts/knob_matrix.sh, ~40 GPU-minutes)
for knob in POISMF_HIP_K50_P32=0 POISMF_HIP_NO_TX POISMF_HIP_TX_MAX=48 POISMF_HIP_K100_MID=1 POISMF_HIP_K100_LANE_MAX=128 POISMF_HIP_K100_LANE_B=0 \
            POISMF_HIP_NO_GIANT_TEAMS POISMF_HIP_NO_LANE_TEAMS POISMF_HIP_LANE_TEAM_STREAM=0 POISMF_HIP_GIANT_TEAMS=2 POISMF_HIP_NO_ROW_INTERRUPT POISMF_HIP_DEVICE_CACHE_MB=4096 POISMF_HIP_NO_TEAM POISMF_HIP_STATIC_ROWS; do
  case $knob in *=*) spec=$knob ;; *) spec=$knob=1 ;; esac
  echo "== $spec"
  env $spec timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_regtile.py tests/test_gpu_rows.py tests/test_gpu_giant.py tests/test_gpu_decisions.py -m gpu -q -x \
      --deselect tests/test_gpu_parity.py::test_sigint_ends_a_half_sweep_within_a_row 2>&1 | tail -1
done
