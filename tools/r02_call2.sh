#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/c2
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_fastmath.py tests/test_gpu_stepper.py tests/test_gpu_multirank.py -x -q -s > gpurun_out/c2/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/c2/status
grep -E "fast fp64|passed|failed|Error" gpurun_out/c2/pytest.log | tail -12
timeout -k 10 300 python tools/sweep.py --lattice D3Q27 --collision KBC --policy FP64FP32 --size 384 --workload periodic --rounds 3 --steps 20 \
  --variant exact_math=1 --variant exact_math=0 --variant "exact_math=0,vec=1" --variant "exact_math=1,vec=1" > gpurun_out/c2/sweep_kbc64.txt 2>&1; echo "sweep rc=$?" | tee -a gpurun_out/c2/status
cat gpurun_out/c2/sweep_kbc64.txt
timeout -k 10 300 python tools/sweep.py --lattice D3Q27 --collision KBC --policy FP64FP64 --size 256 --workload periodic --rounds 3 --steps 20 \
  --variant exact_math=1 --variant exact_math=0 > gpurun_out/c2/sweep_kbc6464.txt 2>&1; echo "sweep2 rc=$?" | tee -a gpurun_out/c2/status
cat gpurun_out/c2/sweep_kbc6464.txt
