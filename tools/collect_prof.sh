#!/bin/bash
# tools/collect_prof.sh <gpurun_out/prof_TAG> <profiles/rNN/NAME>: the judged artefacts of one tools/profile.sh run
D=$1; N=$2
cp $D/summary.md ${N}_summary.md
f=$(find $D/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f ${N}_kernel_stats.csv
grep -h '^{"metric"' $D/kt.log | tail -1 > ${N}_bench_under_rocprof.json
