source tools/gpu_steps.sh
step r3_slab_cost 600 python tools/sweep.py --workload cavity_halfway --size 512 --halo 2 --rounds 3 --steps 40 --variant overlap=1 --variant halo_skip=1 --variant overlap=0
step r3_noslab 300 python tools/sweep.py --workload cavity_halfway --size 512 --rounds 3 --steps 40 --variant fuse2=1
step r3_strong_n1 900 python bench.py --gpus 1 --global-shape 4096x512x512 --steps 20 --warmup 4 --cpu-baseline-seconds 0
