#!/bin/bash
# usage (on the GPU box): scripts/knob_matrix.sh -- the GPU parity suites once under every testing knob of INTEGRATION.md section 5, so that the
# code paths the defaults no longer take (LDS engine for short rows, compact gathers, generic slot counts, no teams, ...) stay green.
# Exactly the knobs that list names (tests/test_knobs.py holds the three -- this script, the document and the sources -- to one another).
for knob in POISMF_HIP_NO_PAD POISMF_HIP_NO_REGTILE POISMF_HIP_STATIC_ROWS POISMF_HIP_NO_FORK POISMF_HIP_LONGROW_NNZ=2000000000 POISMF_HIP_NO_TEAM \
            POISMF_HIP_NO_LANE POISMF_HIP_NO_LS_PRUNE POISMF_HIP_NO_STAGED_UPLOAD POISMF_HIP_STAGED_MIN_BYTES=1 \
            POISMF_HIP_NO_UPLOAD_OVERLAP POISMF_HIP_DEVICE_CACHE_MB=4096 POISMF_HIP_NO_GIANT_TEAMS POISMF_HIP_GIANT_NNZ=2048 \
            POISMF_HIP_NO_LANE_TEAMS POISMF_HIP_NO_ROW_INTERRUPT POISMF_HIP_HOST_THREADS=3; do
  case $knob in *=*) spec=$knob ;; *) spec=$knob=1 ;; esac
  echo "== $spec"
  env $spec timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_regtile.py tests/test_gpu_rows.py tests/test_gpu_team.py tests/test_gpu_giant.py tests/test_gpu_decisions.py -m gpu -q -x \
      --deselect tests/test_gpu_parity.py::test_sigint_ends_a_half_sweep_within_a_row 2>&1 | tail -1
done
