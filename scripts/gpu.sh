#!/bin/bash
# scripts/gpu.sh <timeout_s> '<command>' : gpurun with retries while no GPU slot is free (exit 3 = nothing charged)
t=$1; shift
for i in $(seq 1 30); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 60
done
exit 3
