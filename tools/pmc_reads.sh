#!/bin/bash
# Read amplification of the two-step kernel (runs on the GPU box): FETCH_SIZE per k_step2 launch / bytes of the population field,
# for one workload / size and any number of option variants:   tools/pmc_reads.sh <workload> <size> [variant ...]
# (FETCH_SIZE x 2048 B: the calibration on the copy kernel, profiles/r02/*_summary.md)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; N=$2; shift 2
OUT=$ROOT/gpurun_out/pmc_reads
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for v in "${@:-fuse2=2}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${W}_${N}_$i -- python3 $ROOT/tools/sweep.py --workload $W --size $N --rounds 1 --steps 8 --variant "$v" > $OUT/${W}_${N}_$i.log 2>&1 || { tail -5 $OUT/${W}_${N}_$i.log; exit 1; }
  V="$v" D=$OUT/${W}_${N}_$i N=$N python3 - <<'PY'
import csv, glob, os
vals, dur = [], []
for p in glob.glob(os.environ['D'] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'k_step2' in r['Kernel_Name'] and 'clean' not in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            vals.append(float(r['Counter_Value']))
n = int(os.environ['N']); field = n ** 3 * 19 * 4
ms = [l.split() for l in open(os.environ['D'] + '.log') if l.strip() and not l.startswith(('#', 'variant'))]
if vals:
    rd = sum(vals) / len(vals) * 2048
    print(f"{os.environ['V']:40s} reads {rd / 1e9:7.2f} GB / launch = {rd / field:5.3f} x field   ({len(vals)} launches; under the counters: {ms[-1][-5] if ms else '?'} ms/step)")
else:
    print(os.environ['V'], 'no k_step2 launches')
PY
done
