#!/bin/bash
# usage (on the GPU box, through gpurun):  scripts/profile_round.sh <tag>
# Produces gpurun_out/<tag>/: the default bench line, rocprofv3 --kernel-trace --stats of the SAME command, PMC passes
# (FETCH_SIZE and WRITE_SIZE separately, SQ issue / wait counters) of the timed workload, and the two `extra` workloads.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# 1. the default command, plain and under the kernel trace
$B > $OUT/bench_default.log 2>&1; grep '^{"metric"' $OUT/bench_default.log | tail -1 > $OUT/bench_line.json
rocprofv3 --kernel-trace --stats -d $OUT/kt_default -o kt --output-format csv -- $B > $OUT/kt_default.log 2>&1
grep '^{"metric"' $OUT/kt_default.log | tail -1 > $OUT/bench_line_under_rocprof.json
# 2. the timed workload alone (no extras, no CPU leg): kernel trace + counters, 6 sweeps each
W="--no-cpu --no-extra --steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats -d $OUT/kt_pg10 -o kt --output-format csv -- $B $W > $OUT/kt_pg10.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f10 -o pmc --output-format csv -- $B $W > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w10 -o pmc --output-format csv -- $B $W > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_t10 -o pmc --output-format csv -- $B $W > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc_sq10 -o pmc --output-format csv -- $B $W > /dev/null 2>&1
# 3. PG with one update (the bandwidth point) and CG fp64
rocprofv3 --kernel-trace --stats -d $OUT/kt_pg1 -o kt --output-format csv -- $B $W --maxupd 1 > $OUT/kt_pg1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f1 -o pmc --output-format csv -- $B $W --maxupd 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w1 -o pmc --output-format csv -- $B $W --maxupd 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/kt_cg64 -o kt --output-format csv -- $B --no-cpu --no-extra --steps 3 --warmup 1 --method cg --fp64 > $OUT/kt_cg64.log 2>&1
for d in kt_pg10 kt_pg1 kt_cg64; do grep '^{"metric"' $OUT/$d.log | tail -1 > $OUT/${d}_bench_line.json; done
for f in $(find $OUT -name "*counter_collection.csv"); do python3 $R/scripts/pmc_summary.py $f 0 > $(dirname $f)/summary.txt; done
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
python3 $R/scripts/traffic_from_pmc.py $OUT 6 > $OUT/hbm_traffic.json
cat $OUT/hbm_traffic.json
