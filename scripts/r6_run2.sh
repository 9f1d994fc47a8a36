#!/bin/bash
# round 6, second GPU batch: whole GPU suite on the restructured team code, spill probe, PG partial-set instance, LDS counters of CG fp64
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/tests_gpu.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/probes/spill_divergent.hip -o /tmp/spill_divergent 2> $O/spill_build.log && /tmp/spill_divergent > $O/spill.log 2>&1
B="python3 bench.py --no-cpu --no-extra --steps 10 --warmup 2"
$B > $O/pg_default.log 2>&1; cp bench_full.json $O/pg_default.json
POISMF_HIP_PG_P16=0 $B > $O/pg_nop16.log 2>&1; cp bench_full.json $O/pg_nop16.json
cd /tmp && export TMPDIR=/tmp
C="python3 $R/bench.py --no-cpu --no-extra --method cg --fp64 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $O/pmc_lds_cg64 -o pmc --output-format csv -- $C > $O/pmc_lds_cg64.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA -d $O/pmc_inst_cg64 -o pmc --output-format csv -- $C > $O/pmc_inst_cg64.log 2>&1
for f in $(find $O -name "*counter_collection.csv"); do python3 $R/scripts/pmc_summary.py $f 0 > $(dirname $f)/summary.txt; done
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cd $R
tail -4 $O/tests_gpu.log; cat $O/spill.log; for f in $O/pg_*.log; do echo $f; tail -1 $f | cut -c1-300; done
