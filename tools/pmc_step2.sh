#!/bin/bash
# PMC comparison of the two-step kernel on different wall set-ups (runs on the GPU box).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_step2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in periodic one_cell zwalls zwalls_fw; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/$w -- python3 $ROOT/tools/sweep.py --workload $w --rounds 1 --steps 8 --variant "fuse2=2,fuse2_lpt=0" > $OUT/$w.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
out=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/pmc_step2'
for w in ('periodic','one_cell','zwalls','zwalls_fw'):
    acc=defaultdict(lambda: defaultdict(list))
    for p in glob.glob(f'{out}/{w}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if 'k_step2' in r['Kernel_Name']:
                acc[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,cs in acc.items():
        print(w, k, {c: f"{sum(v)/len(v):.3g}" for c,v in sorted(cs.items())})
PY
