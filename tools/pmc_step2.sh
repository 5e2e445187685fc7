#!/bin/bash
# SQ counters of the two-step kernel on different wall set-ups (runs on the GPU box):
#   tools/pmc_step2.sh [workload ...]      default: periodic zwalls
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_step2
mkdir -p $OUT
WL=${@:-periodic zwalls}
cd /tmp && export TMPDIR=/tmp
for w in $WL; do
  for pass in a b; do
    if [ $pass = a ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
    else C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; fi
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${w}_$pass -- python3 $ROOT/tools/sweep.py --workload $w --rounds 1 --steps 8 --variant "fuse2=2" > $OUT/${w}_$pass.log 2>&1 || { tail -5 $OUT/${w}_$pass.log; exit 1; }
  done
done
WL="$WL" python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
out=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/pmc_step2'
for w in os.environ['WL'].split():
    acc=defaultdict(list)
    for p in glob.glob(f'{out}/{w}_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if 'k_step2' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(w, {c: f"{sum(v)/len(v):.4g}" for c,v in sorted(acc.items())})
PY
