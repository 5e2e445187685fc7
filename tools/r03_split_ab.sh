#!/bin/bash
# Round 3: phase A's arithmetic + next pulls before the barrier that ends phase B (XLB_STEP2_SPLIT_BC / _PLAIN) — parity of the variant
# builds, then A/B against the default build (tools/build_variant.sh split_bc1 / split_idle).
source tools/gpu_steps.sh
for v in split_bc1 split_idle; do
  XLBHIP_LIB=$PWD/xlb_amd/lib/$v.so step r3_split_parity_$v 500 python -m pytest tests/test_gpu_stepper.py -x -q -k "two_step or fusion or strips or pair"
  grep -q "passed" gpurun_out/r3_split_parity_$v.log || exit 1
  grep -q "failed" gpurun_out/r3_split_parity_$v.log && exit 1
done
step r3_split_ab2 1000 bash tools/ab_libs.sh "periodic cavity_halfway cavity_fullway" 3 xlb_amd/lib/libxlbhip.so xlb_amd/lib/split_bc1.so xlb_amd/lib/split_idle.so
VARIANT="fuse2=2,fuse2_strips=0" step r3_split_ab2_nostrips 400 bash tools/ab_libs.sh "cavity_halfway" 2 xlb_amd/lib/libxlbhip.so xlb_amd/lib/split_bc1.so xlb_amd/lib/split_idle.so
