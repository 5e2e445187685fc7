#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/c3
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_stepper.py tests/test_gpu_fullsize.py -x -q > gpurun_out/c3/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/c3/status
tail -2 gpurun_out/c3/pytest.log
timeout -k 10 300 python tools/sweep.py --workload cavity_halfway --size 512 --rounds 3 --steps 20 \
  --variant fuse2_lpt=1 --variant fuse2_lpt=3 --variant fuse2_lpt=0 --variant "fuse2_lpt=1,fuse2_xseg=4" --variant "fuse2_lpt=3,fuse2_xseg=4" > gpurun_out/c3/sweep_lpt.txt 2>&1; echo "sweep rc=$?" | tee -a gpurun_out/c3/status
cat gpurun_out/c3/sweep_lpt.txt
