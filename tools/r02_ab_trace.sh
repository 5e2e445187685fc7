#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/abtrace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for L in ab_slots3 libxlbhip; do
  XLBHIP_LIB=$ROOT/xlb_amd/lib/$L.so rocprofv3 --kernel-trace --output-format csv -d $OUT/${L}_$rep -- python3 $ROOT/tools/sweep.py --workload periodic --rounds 1 --steps 20 --variant fuse2=1 > $OUT/${L}_$rep.log 2>&1
  python3 - <<PY
import csv,glob
rows=[r for p in glob.glob("$OUT/${L}_$rep/**/*kernel_trace.csv",recursive=True) for r in csv.DictReader(open(p)) if "k_step2" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
r=rows[-1]
print("$L rep$rep", len(d), "launches avg us", sum(d)/len(d), "min", min(d), {k:r[k] for k in r if k in ("Grid_Size_X","Workgroup_Size_X","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count")})
PY
  grep "^fuse2" $OUT/${L}_$rep.log
done; done
