#!/bin/bash
# round-2 GPU call 1: tests, bench (1 GPU + 2-rank rehearsal from a plain shell), configs[4] profiles
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/c1
cd $ROOT
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/c1/status
tail -3 gpurun_out/c1/pytest.log
timeout -k 10 200 python bench.py --steps 60 --warmup 10 > gpurun_out/c1/bench1.json 2> gpurun_out/c1/bench1.err; echo "bench1 rc=$?" | tee -a gpurun_out/c1/status
cat gpurun_out/c1/bench1.json
XLB_BENCH_TRANSPORT=host timeout -k 10 200 python bench.py --gpus 2 --size 64 --steps 10 --warmup 2 > gpurun_out/c1/bench2host.json 2> gpurun_out/c1/bench2host.err; echo "bench2host rc=$?" | tee -a gpurun_out/c1/status
cat gpurun_out/c1/bench2host.json
XLB_BENCH_TRANSPORT=host timeout -k 10 200 python bench.py --gpus 2 --global-shape 128x64x64 --steps 10 --warmup 2 --cpu-baseline-seconds 0 > gpurun_out/c1/bench2strong.json 2> gpurun_out/c1/bench2strong.err; echo "bench2strong rc=$?" | tee -a gpurun_out/c1/status
cat gpurun_out/c1/bench2strong.json
timeout -k 10 300 bash tools/profile.sh r02_d3q27_kbc_384_fp32 --workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP32FP32 --omega 1.9 --steps 100 > gpurun_out/c1/prof_fp32.log 2>&1; echo "prof fp32 rc=$?" | tee -a gpurun_out/c1/status
timeout -k 10 300 bash tools/profile.sh r02_d3q27_kbc_384_fp64fp32 --workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9 --steps 100 > gpurun_out/c1/prof_fp64.log 2>&1; echo "prof fp64 rc=$?" | tee -a gpurun_out/c1/status
tail -30 gpurun_out/c1/prof_fp64.log
