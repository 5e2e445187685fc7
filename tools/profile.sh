#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the
# bench workload, and the counter calibration on a copy kernel of known traffic.
# Usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
# a tag re-used across runs must not keep an earlier run's CSVs (a kernel_stats.csv of an older build ended up
# beside a newer summary): every run starts from an empty directory
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel trace: the default bench command (200 steps, 10 warm-up) minus the CPU-baseline leg; counters: a shorter run
BENCH="python3 $ROOT/bench.py --steps 30 --warmup 6 --cpu-baseline-seconds 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py --cpu-baseline-seconds 0 $* > $OUT/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
# calibration: copy kernels with known byte counts (10.2 GB each way), same counters
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- python3 $ROOT/tools/copy_bw.py 512 3 > $OUT/cal_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- python3 $ROOT/tools/copy_bw.py 512 3 > $OUT/cal_write.log 2>&1
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.md 2>&1 || true
# the stats file of THIS run (the directory was empty when it started), for tools/collect_prof.sh: gpurun merges a run's files INTO the
# caller's gpurun_out/, where an earlier run's files of the same tag may still lie
(cd $OUT && find kt -name "*kernel_stats.csv") > $OUT/kt_stats_of_this_run.txt
find $OUT -name '*.csv' -size +2M -delete
cat $OUT/summary.md
