#!/bin/bash
# usage (on the GPU box): scripts/probes/abi_timeline.sh [threads ...] -- run_poismf on the 1e8-nnz matrix with the library's own
# phase stamps (POISMF_HIP_VERBOSE=2), once per host-thread count
for t in "${@:-8}"; do
  echo "== POISMF_HIP_HOST_THREADS=$t"
  POISMF_HIP_VERBOSE=2 POISMF_HIP_HOST_THREADS=$t python3 scripts/time_abi.py C4 2>&1 | grep -v amdgpu.ids
done
