#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python examples/cavity_3d_reference_loop_hip.py 256 100 2>/dev/null | tail -2
XLB_BENCH_TRANSPORT=host timeout -k 10 200 python bench.py --gpus 2 --size 64 --steps 10 --warmup 2 --cpu-baseline-seconds 0 2>/dev/null | cut -c1-400
bash tools/r02_final_prof.sh
python tools/design_table.py > gpurun_out/design_table.md 2> gpurun_out/design_table.err; cat gpurun_out/design_table.md
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-900 gpurun_out/bench_default.json
