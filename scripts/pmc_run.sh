#!/bin/bash
# usage: scripts/pmc_run.sh <outdir under gpurun_out> <bench args...>   (run on the GPU box through gpurun)
# One rocprofv3 --pmc pass with SQ issue/wait counters, one with TCC hit/miss, one kernel-trace pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $OUT/sq -o sq --output-format csv -- python3 $R/bench.py --no-cpu --no-extra "$@" > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/sq2 -o sq2 --output-format csv -- python3 $R/bench.py --no-cpu --no-extra "$@" > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu --no-extra "$@" > $OUT/kt.log 2>&1
for f in $(find $OUT -name "*counter_collection.csv"); do python3 $R/scripts/pmc_summary.py $f 1000 > ${f%.csv}.summary.txt; done
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -30
