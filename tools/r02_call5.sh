#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/c5
cd $ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_stepper.py -x -q > gpurun_out/c5/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/c5/status
tail -5 gpurun_out/c5/pytest.log
[ $rc -eq 0 ] || exit 1
S="python tools/sweep.py --size 512 --rounds 3 --steps 20"
timeout -k 10 300 $S --workload cavity_halfway --variant fuse2_tile=0 --variant fuse2_tile=2 > gpurun_out/c5/sweep_h.txt 2>&1; cat gpurun_out/c5/sweep_h.txt
timeout -k 10 300 $S --workload periodic --variant fuse2_tile=0 --variant fuse2_tile=2 --variant fuse2=0 > gpurun_out/c5/sweep_p.txt 2>&1; cat gpurun_out/c5/sweep_p.txt
timeout -k 10 300 python tools/sweep.py --size 384 --rounds 3 --steps 20 --lattice D3Q27 --workload periodic --variant fuse2=0 --variant fuse2=1 --variant "fuse2=2,fuse2_xseg=4" --variant "fuse2=2,fuse2_xseg=2" > gpurun_out/c5/sweep_d3q27.txt 2>&1; cat gpurun_out/c5/sweep_d3q27.txt
