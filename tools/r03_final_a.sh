#!/bin/bash
# round-3 evidence, part A (one gpurun call): the whole GPU suite, smoke, the default bench line, the reference-protocol loop,
# rehearsals of the N > 1 path on one GPU, the design table
source tools/gpu_steps.sh
step r3_final_tests 1100 python -m pytest tests -m gpu -x -q
step r3_final_smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step r3_final_bench_default 400 python bench.py
step r3_final_refloop 400 python tools/ref_protocol_bench.py
XLB_BENCH_TRANSPORT=ipc step r3_final_ipc2_256 300 python bench.py --gpus 2 --size 256 --steps 100 --cpu-baseline-seconds 0
XLB_BENCH_TRANSPORT=ipc step r3_final_ipc2_256_skip 300 python bench.py --gpus 2 --size 256 --steps 100 --cpu-baseline-seconds 0 --opt halo_skip=1
step r3_final_auto2_256 300 python bench.py --gpus 2 --size 256 --steps 40 --cpu-baseline-seconds 0
step r3_final_design_table 1100 python tools/design_table.py
