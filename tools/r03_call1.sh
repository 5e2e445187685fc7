source tools/gpu_steps.sh
step r3_t1_multirank 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu
step r3_t1_lazy 400 python -m pytest tests/test_gpu_lazy_pairs.py tests/test_gpu_postprocess.py -x -q -m gpu
step r3_t1_fullsize 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c_oracle"
XLB_BENCH_TRANSPORT=ipc step r3_b_ipc2_256 400 python bench.py --gpus 2 --size 256 --steps 100
XLB_BENCH_TRANSPORT=ipc step r3_b_ipc2_256_skip 400 python bench.py --gpus 2 --size 256 --steps 100 --opt halo_skip=1
step r3_b_default 400 python bench.py
