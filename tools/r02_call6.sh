#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/c6
XLBHIP_LIB=$ROOT/xlb_amd/lib/ab_packed.so timeout -k 10 300 python -m pytest tests/test_gpu_stepper.py -x -q 2>&1 | tail -2
XLBHIP_LIB=$ROOT/xlb_amd/lib/ab_unpacked.so timeout -k 10 300 python -m pytest tests/test_gpu_stepper.py -x -q 2>&1 | tail -2
VARIANT=fuse2=1 bash tools/ab_libs.sh "periodic cavity_halfway" 2 xlb_amd/lib/ab_slots3.so xlb_amd/lib/ab_unpacked.so xlb_amd/lib/ab_packed.so | tee gpurun_out/c6/ab.txt
SWEEP_ARGS="--size 384 --lattice D3Q27" VARIANT=fuse2=1 bash tools/ab_libs.sh "periodic" 2 xlb_amd/lib/libxlbhip.so xlb_amd/lib/ab_packed.so | tee gpurun_out/c6/ab27.txt
