source tools/gpu_steps.sh
step r3_t8_stepper 900 python -m pytest tests/test_gpu_stepper.py tests/test_gpu_fullsize.py tests/test_gpu_lazy_pairs.py -x -q -m gpu
step r3_d3q27_walls_sweep 600 python tools/sweep.py --workload cavity_halfway --size 384 --lattice D3Q27 --rounds 3 --steps 40 --variant fuse2=0 --variant fuse2=1 --variant fuse2=2
step r3_d3q27_walls_sweep_fw 600 python tools/sweep.py --workload cavity_fullway --size 384 --lattice D3Q27 --rounds 3 --steps 40 --variant fuse2=0 --variant fuse2=1
step r3_slab_strips_halfway 600 python tools/sweep.py --workload cavity_halfway --size 512 --halo 2 --rounds 4 --steps 40 --variant fuse2_strips=0 --variant fuse2_strips=1
step r3_slab_periodic 600 python tools/sweep.py --workload periodic --size 512 --halo 2 --rounds 3 --steps 40 --variant fuse2_strips=0 --variant fuse2_strips=2
step r3_kbc_auto 600 python tools/sweep.py --workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --rounds 3 --steps 40 --variant fuse2=0 --variant fuse2=1
