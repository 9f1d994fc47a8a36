#!/bin/bash
# usage (on the GPU box, through gpurun):  scripts/profile_round.sh <tag> [round name stamped into hbm_traffic.json, default r06]
# Produces gpurun_out/<tag>/: the default bench line, rocprofv3 --kernel-trace --stats of the workloads behind it (PG fp32
# with maxupd 10 and 1, CG fp64, CG fp32, TNCG fp32, all on the 1M x 100K / 1e8-nnz matrix) and PMC passes (FETCH_SIZE and WRITE_SIZE separately,
# TCC hit / miss, SQ issue / wait counters) of the same commands.  scripts/install_profiles.sh copies the summaries to profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# 1. the default command as the driver runs it
$B > $OUT/bench_default.log 2>&1; grep '^{"metric"' $OUT/bench_default.log | tail -1 > $OUT/bench_line.json; cp $R/bench_full.json $OUT/bench_full.json
# 2. one workload per command: kernel trace, then counters.  Each run makes warmup + steps timed sweeps and the same again
#    with the session's own event timing on (the pass `roofline` is computed from): 2 x (1 + STEPS) sweeps per run.
run() {   # name, sweeps flags..., bench flags
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt --output-format csv -- $B --no-cpu --no-extra "$@" > $OUT/kt_$name.log 2>&1
  cp $R/bench_full.json $OUT/kt_${name}_bench_line.json   # (the full object of that run: bench.py's printed line is the compact one)
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f_$name -o pmc --output-format csv -- $B --no-cpu --no-extra "$@" > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w_$name -o pmc --output-format csv -- $B --no-cpu --no-extra "$@" > /dev/null 2>&1
  # (L2 hit / miss: the headline workload only -- the whole round has to fit one gpurun call)
  [ $name = pg10 ] && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_t_$name -o pmc --output-format csv -- $B --no-cpu --no-extra "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc_sq_$name -o pmc --output-format csv -- $B --no-cpu --no-extra "$@" > /dev/null 2>&1
}
run pg10 --steps 5 --warmup 1
run pg1 --steps 5 --warmup 1 --maxupd 1
run cg64 --steps 2 --warmup 1 --method cg --fp64
run cg32 --steps 2 --warmup 1 --method cg
run tncg32 --steps 2 --warmup 1 --method tncg
summaries() {
  for f in $(find $OUT -name "*counter_collection.csv"); do python3 $R/scripts/pmc_summary.py $f 0 > $(dirname $f)/summary.txt; done
  find $OUT -name "*counter_collection.csv" -delete
  find $OUT -name "*kernel_trace.csv" -delete
  find $OUT -name "*agent_info.csv" -delete
  python3 $R/scripts/traffic_from_pmc.py $OUT ${2:-r06} > $OUT/hbm_traffic.json
}
summaries "$@"
# config C5 (its own matrix, k = 100, tncg fp64) through scripts/run_config.py: 2 warm-up + 3 timed sweeps, no oracle sample.  Last, and
# each counter pass under its own time limit: with counters on, dispatches are serialised, and this workload's B half forks its giant
# rows onto a second stream and holds the other bins back until they are on the chip -- with hipStreamWaitValue32 (until mid round 4) the
# first of these passes never ended.  The hold-back is a bounded gate kernel now; the limits stay.
C5="python3 $R/scripts/run_config.py C5 --warmup 2 --sweeps 3 --sample 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_c5 -o kt --output-format csv -- $C5 > $OUT/kt_c5.log 2>&1
grep '^{"config"' $OUT/kt_c5.log | tail -1 > $OUT/kt_c5_run_config.json
# the TIMELINE of that run's last sweep (round 6): every half_sweep_* / team_* / hold_back dispatch with its start, end and queue -- which launch
# ran beside which on the two streams of the B half (the --stats table cannot say)
python3 $R/scripts/trace_timeline.py $OUT/kt_c5 60 > $OUT/kt_c5_timeline.txt 2>&1
timeout 420 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f_c5 -o pmc --output-format csv -- $C5 > $OUT/pmc_f_c5.log 2>&1 || echo "pmc_f_c5: rc $?" >> $OUT/c5_pmc_status.txt
timeout 420 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w_c5 -o pmc --output-format csv -- $C5 > $OUT/pmc_w_c5.log 2>&1 || echo "pmc_w_c5: rc $?" >> $OUT/c5_pmc_status.txt
timeout 420 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc_sq_c5 -o pmc --output-format csv -- $C5 > $OUT/pmc_sq_c5.log 2>&1 || echo "pmc_sq_c5: rc $?" >> $OUT/c5_pmc_status.txt
summaries "$@"
cat $OUT/hbm_traffic.json
