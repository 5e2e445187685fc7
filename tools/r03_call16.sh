source tools/gpu_steps.sh
step r3_t16 1000 python -m pytest tests/test_gpu_stepper.py tests/test_gpu_multirank.py tests/test_gpu_lazy_pairs.py tests/test_gpu_fullsize.py -x -q -m gpu
step r3_driver_style 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-seconds 0
step r3_driver_style_nostrips 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-seconds 0 --opt fuse2_strips=0
step r3_driver_style2 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-seconds 0
step r3_driver_style_nostrips2 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-seconds 0 --opt fuse2_strips=0
