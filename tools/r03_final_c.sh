#!/bin/bash
# round-3 evidence, part C: rehearsals of the N > 1 path on one GPU (the pool's process guard allows 6 processes per card)
source tools/gpu_steps.sh
step r3_final_auto2_256 300 python bench.py --gpus 2 --size 256 --steps 40 --cpu-baseline-seconds 0
XLB_BENCH_TRANSPORT=ipc step r3_final_ipc5_512 900 python bench.py --gpus 5 --size 512 --steps 12 --warmup 4 --cpu-baseline-seconds 0
XLB_BENCH_TRANSPORT=ipc step r3_final_ipc5_512_skip 900 python bench.py --gpus 5 --size 512 --steps 12 --warmup 4 --cpu-baseline-seconds 0 --opt halo_skip=1
