#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/c4
cd $ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_stepper.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_inlet_outlet.py -x -q > gpurun_out/c4/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/c4/status
tail -5 gpurun_out/c4/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/sweep.py --workload cavity_halfway --size 512 --rounds 3 --steps 20 --variant fuse2_lpt=1 > gpurun_out/c4/sweep_h.txt 2>&1; echo "sweep rc=$?" | tee -a gpurun_out/c4/status
cat gpurun_out/c4/sweep_h.txt
timeout -k 10 300 python tools/sweep.py --workload periodic --size 512 --rounds 3 --steps 20 --variant fuse2=1 > gpurun_out/c4/sweep_p.txt 2>&1; echo "sweep rc=$?" | tee -a gpurun_out/c4/status
cat gpurun_out/c4/sweep_p.txt
