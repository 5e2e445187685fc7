source tools/gpu_steps.sh
step r3_strips_sweep_halfway 600 python tools/sweep.py --workload cavity_halfway --size 512 --rounds 6 --steps 40 --variant fuse2_strips=0 --variant fuse2_strips=1
step r3_strips_sweep_fullway 600 python tools/sweep.py --workload cavity_fullway --size 512 --rounds 6 --steps 40 --variant fuse2_strips=0 --variant fuse2_strips=1
step r3_strips_reads_halfway 600 bash tools/pmc_reads.sh cavity_halfway 512 fuse2_strips=0 fuse2_strips=1
step r3_strips_reads_periodic 600 bash tools/pmc_reads.sh periodic 512 fuse2_strips=0 fuse2_strips=1
XLBHIP_LIB=$PWD/xlb_amd/lib/r2kernel.so step r3_r2kernel_reads_periodic 400 bash tools/pmc_reads.sh periodic 512 fuse2_strips=0
