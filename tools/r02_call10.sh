#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/c10
timeout -k 10 400 python -m pytest tests/test_gpu_stepper.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_inlet_outlet.py -x -q > gpurun_out/c10/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/c10/pytest.log
[ $rc -eq 0 ] || exit 1
S="python tools/sweep.py --size 512 --rounds 3 --steps 20"
$S --workload cavity_halfway --variant fuse2_shift=1 --variant fuse2_shift=0 --variant "fuse2_shift=1,fuse2_lpt=0" 2>/dev/null | tee gpurun_out/c10/sweep_h.txt
$S --workload cavity_fullway --variant fuse2_shift=1 --variant fuse2_shift=0 2>/dev/null | tee gpurun_out/c10/sweep_f.txt
$S --workload periodic --variant fuse2=1 2>/dev/null | tee gpurun_out/c10/sweep_p.txt
python tools/sweep.py --size 256 --rounds 3 --steps 40 --workload cavity_halfway --variant fuse2_shift=1 --variant fuse2_shift=0 --variant fuse2=0 2>/dev/null | tee gpurun_out/c10/sweep_h256.txt
