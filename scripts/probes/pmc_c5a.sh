#!/bin/bash
# usage (GPU box): scripts/probes/pmc_c5a.sh <variant dir under _ab>   -- issue / wait / instruction-cache counters of config C5's A half (scripts/probes/c5_ahalf.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/_ab/$1
OUT=$R/gpurun_out/pmc_c5a_$1; mkdir -p $OUT
python3 $R/_ab/base/scripts/probes/c5_ahalf.py save
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/a -o a --output-format csv -- python3 $V/scripts/probes/c5_ahalf.py > $OUT/a.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/b -o b --output-format csv -- python3 $V/scripts/probes/c5_ahalf.py > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC -d $OUT/c -o c --output-format csv -- python3 $V/scripts/probes/c5_ahalf.py > $OUT/c.log 2>&1
for f in $(find $OUT -name "*counter_collection.csv"); do python3 $R/scripts/pmc_summary.py $f 1000; done
find $OUT -name "*.csv" -delete
