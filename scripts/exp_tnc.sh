#!/bin/bash
# usage (GPU box): scripts/exp_tnc.sh <tag> <variant dir under _ab | .> ...   -- config C5 (scripts/probes/c5_halves.py) and TNCG fp32 on the metric's
# matrix (scripts/probes/c4_halves.py), once per build: per-launch times, the solvers' evaluation counts, a hash of the factors -- same-box A/B of
# TNC changes (bit-identical builds print the same hashes).  EXP_TNC_NO_F32=1 / EXP_TNC_NO_C5=1 skip a leg.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
for v in "$@"; do
  d=$R; name=main
  if [ "$v" != "." ]; then d=$R/_ab/$v; name=$v; fi
  for s in c5_halves.py c4_halves.py; do cp $R/scripts/probes/$s $d/scripts/probes/$s 2>/dev/null; done
  if [ -z "$EXP_TNC_NO_C5" ]; then echo "== $name: C5 tncg fp64 k=100"; python3 $d/scripts/probes/c5_halves.py 2 3 2>&1 | tee $OUT/c5_$name.log; fi
  if [ -z "$EXP_TNC_NO_F32" ]; then echo "== $name: tncg fp32, C4 matrix"; python3 $d/scripts/probes/c4_halves.py tncg 1 2 3 2>&1 | tee $OUT/t32_$name.log; fi
done
