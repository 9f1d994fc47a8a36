#!/bin/bash
# usage: scripts/resource_usage.sh [out file, default profiles/r06/kernel_resource_usage.txt]
# Every row kernel's VGPRs / AGPRs / scratch / waves per SIMD / LDS as the compiler reports them (-Rpass-analysis=kernel-resource-usage),
# one device-only compile of poismf_hip.hip per solver translation unit and precision, with the product build's flags.  ~6 minutes on 8 cores.
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r06/kernel_resource_usage.txt}
TMP=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -Wno-unused-const-variable -Wno-pass-failed $POISMF_HIP_EXTRA_FLAGS"
for prec in f64 f32; do
  for tu in 1:tncg 2:cg 3:pg; do
    n=${tu%%:*}; name=${tu##*:}
    pf=""; [ $prec = f32 ] && pf="-DUSE_FLOAT"
    ( /opt/rocm/bin/hipcc $FLAGS $pf -DPMF_TU=$n --cuda-device-only -c -Rpass-analysis=kernel-resource-usage -o /dev/null poismf_amd/csrc/poismf_hip.hip > $TMP/$prec.$name.txt 2>&1 ) &
  done
done
wait
{
  echo "# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage of poismf_hip.hip, one translation unit per solver and precision (scripts/resource_usage.sh)"
  echo "# precision solver | VGPRs AGPRs scratch[B/lane] waves/SIMD LDS[B/workgroup] | kernel"
  for prec in f32 f64; do for name in cg pg tncg; do
    python3 scripts/res_usage.py $TMP/$prec.$name.txt | sort -t'|' -k2 | while IFS= read -r line; do printf "%s %-4s | %s\n" $prec $name "$line"; done
  done; done
} > $OUT
rm -rf $TMP
wc -l $OUT
