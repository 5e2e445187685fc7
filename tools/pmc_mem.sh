#!/bin/bash
# Memory-path counters of the two-step kernel next to the copy yardstick (runs on the GPU box):  tools/pmc_mem.sh [workload] [size]
# Averages per launch; LEVEL / REQ = mean latency in L2 clocks of a read from the fabric.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-periodic}; N=${2:-512}
OUT=$ROOT/gpurun_out/pmc_mem
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr"
P2="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum"
P3="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum"
# (TA_* + TCP_* in one pass "exceeds the capabilities of the hardware": rocprofv3 aborts and hangs — one block per pass, and a timeout)
P4="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
P5="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
P6="TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/s$i -- python3 $ROOT/tools/sweep.py --workload $W --size $N --rounds 1 --steps 8 --variant "fuse2=2" > $OUT/s$i.log 2>&1 || { tail -5 $OUT/s$i.log; exit 1; }
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/c$i -- python3 $ROOT/tools/copy_bw.py $N 3 > $OUT/c$i.log 2>&1 || { tail -5 $OUT/c$i.log; exit 1; }
done
OUT=$OUT python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
out = os.environ['OUT']
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        k = r['Kernel_Name']
        name = 'k_step2' if ('k_step2' in k and 'clean' not in k) else ('k_copy' if 'k_copy' in k else None)
        if name:
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name in acc:
    print(name)
    for c, v in sorted(acc[name].items()):
        print(f"  {c:42s} {sum(v) / len(v):14.5g}   ({len(v)} launches)")
PY
