#!/bin/bash
# tools/collect_prof.sh <gpurun_out/prof_TAG> <profiles/rNN/NAME>: the judged artefacts of one tools/profile.sh run.
# The kernel-stats CSV must be THE one of the kernel-trace run the summary was computed from.  profile.sh wipes its output
# directory on the GPU box and records the name of the run's stats file (kt_stats_of_this_run.txt); gpurun then MERGES the files into
# the local gpurun_out/, where files of an earlier run of the same tag may remain — those are ignored here.  Anything ambiguous
# is an error instead of a guess.
set -e
D=$1; N=$2
if [ -s $D/kt_stats_of_this_run.txt ]; then
  mapfile -t stats < <(sed "s#^#$D/#" $D/kt_stats_of_this_run.txt)
else
  mapfile -t stats < <(find $D/kt -name "*kernel_stats.csv")
fi
if [ ${#stats[@]} -ne 1 ] || [ ! -f "${stats[0]}" ]; then
  echo "collect_prof.sh: expected exactly one *kernel_stats.csv of the run under $D/kt, found ${#stats[@]}" >&2
  exit 1
fi
cp $D/summary.md ${N}_summary.md
cp ${stats[0]} ${N}_kernel_stats.csv
grep -h '^{"metric"' $D/kt.log | tail -1 > ${N}_bench_under_rocprof.json
# the kernel the summary names first must be in the CSV too (same instantiation, same run), with the same number of calls
k=$(grep -m1 -o '`k_step[^`]*`' ${N}_summary.md | tr -d '`' | cut -c1-60)
sed 's/xlb:://g; s/void //g' ${N}_kernel_stats.csv | grep -q -F "$k" || { echo "collect_prof.sh: '$k' of the summary is not in the kernel-stats CSV" >&2; exit 1; }
