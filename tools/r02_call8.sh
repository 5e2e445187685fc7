#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/c8
timeout -k 10 400 python -m pytest tests/test_gpu_stepper.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_inlet_outlet.py -x -q > gpurun_out/c8/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/c8/pytest.log
[ $rc -eq 0 ] || exit 1
S="python tools/sweep.py --size 512 --rounds 3 --steps 20"
$S --workload cavity_halfway --variant fuse2_clean=1 --variant fuse2_clean=0 --variant "fuse2_clean=1,fuse2_xseg=4" --variant "fuse2_clean=1,fuse2_xseg=16" 2>/dev/null | tee gpurun_out/c8/sweep_h.txt
$S --workload cavity_fullway --variant fuse2_clean=1 --variant fuse2_clean=0 --variant "fuse2_clean=1,fuse2_xseg=8" --variant "fuse2_clean=1,fuse2_lpt=2,fuse2_xseg=8" 2>/dev/null | tee gpurun_out/c8/sweep_f.txt
$S --workload periodic --variant fuse2=1 2>/dev/null | tee gpurun_out/c8/sweep_p.txt
