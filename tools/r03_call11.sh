source tools/gpu_steps.sh
SWEEP_ARGS="--size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9" VARIANT="fuse2=0" step r3_kbc2_single 600 bash tools/ab_libs.sh periodic 3 xlb_amd/lib/gamma64.so xlb_amd/lib/libxlbhip.so
SWEEP_ARGS="--size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9" VARIANT="fuse2=2" step r3_kbc2_pairs 600 bash tools/ab_libs.sh periodic 3 xlb_amd/lib/gamma64.so xlb_amd/lib/libxlbhip.so
step r3_kbc2_sweep 600 python tools/sweep.py --workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9 --rounds 4 --steps 40 --variant fuse2=0 --variant fuse2=2
step r3_periodic_sweep 600 python tools/sweep.py --workload periodic --size 512 --rounds 3 --steps 40 --variant fuse2=1 --variant fuse2_strips=2 --variant fuse2=0
