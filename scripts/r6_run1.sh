#!/bin/bash
# round 6, first GPU batch: team bookkeeping restructured, partial set of 48, tolerances, spill probe
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_giant.py tests/test_gpu_team.py -x -q -s 2>&1 | tail -40 > $O/tests_team.log
python -m pytest tests/test_gpu_decisions.py tests/test_gpu_regtile.py -x -q 2>&1 | tail -5 > $O/tests_dec.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/probes/spill_divergent.hip -o /tmp/spill_divergent && /tmp/spill_divergent > $O/spill.log 2>&1
python scripts/run_config.py C5 --warmup 2 --sweeps 3 --sample 0 > $O/c5.log 2>&1
B="python bench.py --no-cpu --no-extra --method cg --fp64 --steps 3 --warmup 1"
$B > $O/cg64_default.log 2>&1; cp bench_full.json $O/cg64_default.json
POISMF_HIP_K50_P48=0 $B > $O/cg64_nop48.log 2>&1; cp bench_full.json $O/cg64_nop48.json
POISMF_HIP_K50_MID=2 $B > $O/cg64_mid2.log 2>&1; cp bench_full.json $O/cg64_mid2.json
POISMF_HIP_K50_MID=2 POISMF_HIP_K50_P48=0 $B > $O/cg64_mid2_nop48.log 2>&1; cp bench_full.json $O/cg64_mid2_nop48.json
tail -3 $O/tests_team.log $O/tests_dec.log $O/spill.log; grep -h '"config"' $O/c5.log | cut -c1-600; for f in $O/cg64_*.log; do echo $f; tail -1 $f | cut -c1-400; done
