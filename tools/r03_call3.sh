source tools/gpu_steps.sh
step r3_t3_stepper 900 python -m pytest tests/test_gpu_stepper.py tests/test_gpu_lazy_pairs.py -x -q -m gpu
step r3_t3_multirank 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu
step r3_t3_fastmath 600 python -m pytest tests/test_gpu_fastmath.py tests/test_gpu_fullsize.py -x -q -m gpu -s -k "fast or config5 or kbc or c_oracle"
step r3_strips_ab_cavity 600 python tools/sweep.py --workload cavity_halfway --size 512 --rounds 3 --steps 40 --variant fuse2_strips=0 --variant fuse2_strips=1
step r3_strips_ab_periodic 600 python tools/sweep.py --workload periodic --size 512 --rounds 3 --steps 40 --variant fuse2_strips=0 --variant fuse2_strips=1
step r3_strips_ab_fullway 600 python tools/sweep.py --workload cavity_fullway --size 512 --rounds 2 --steps 40 --variant fuse2_strips=0 --variant fuse2_strips=1
SWEEP_ARGS="--size 384 --lattice D3Q27 --collision KBC --policy FP64FP32" VARIANT="fuse2=0" step r3_kbc_gamma_ab 600 bash tools/ab_libs.sh periodic 3 xlb_amd/lib/gamma64.so xlb_amd/lib/libxlbhip.so
SWEEP_ARGS="--size 384 --lattice D3Q27 --collision KBC --policy FP64FP32" VARIANT="fuse2=2" step r3_kbc_gamma_ab_step2 600 bash tools/ab_libs.sh periodic 2 xlb_amd/lib/gamma64.so xlb_amd/lib/libxlbhip.so
XLB_BENCH_TRANSPORT=ipc step r3_b3_ipc2_512_kernel 400 python bench.py --gpus 2 --size 512 --steps 60
XLB_BENCH_TRANSPORT=ipc step r3_b3_ipc3_256_kernel 400 python bench.py --gpus 3 --size 256 --steps 60 --workload cavity_fullway
