source tools/gpu_steps.sh
step r3_t2_multirank 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu
XLB_BENCH_TRANSPORT=ipc step r3_b2_ipc2_256 400 python bench.py --gpus 2 --size 256 --steps 100
XLB_BENCH_TRANSPORT=ipc step r3_b2_ipc2_256_kernel 400 python bench.py --gpus 2 --size 256 --steps 100 --opt ipc_copy=1
XLB_BENCH_TRANSPORT=ipc step r3_b2_ipc2_256_skip 400 python bench.py --gpus 2 --size 256 --steps 100 --opt halo_skip=1
XLB_BENCH_TRANSPORT=ipc step r3_b2_ipc2_512 400 python bench.py --gpus 2 --size 512 --steps 60
XLB_BENCH_TRANSPORT=ipc step r3_b2_ipc2_512_skip 400 python bench.py --gpus 2 --size 512 --steps 60 --opt halo_skip=1
# what-if builds of the two-step kernel: halo columns free (1), halo columns + rows free (3) — timing only
step r3_whatif_ab 900 bash tools/ab_libs.sh "periodic cavity_halfway" 2 xlb_amd/lib/libxlbhip.so xlb_amd/lib/whatif1.so xlb_amd/lib/whatif3.so
XLBHIP_LIB=$PWD/xlb_amd/lib/libxlbhip.so step r3_whatif_reads0 400 bash tools/pmc_reads.sh cavity_halfway 512 fuse2=2
XLBHIP_LIB=$PWD/xlb_amd/lib/whatif1.so step r3_whatif_reads1 400 bash tools/pmc_reads.sh cavity_halfway 512 fuse2=2
XLBHIP_LIB=$PWD/xlb_amd/lib/whatif3.so step r3_whatif_reads3 400 bash tools/pmc_reads.sh cavity_halfway 512 fuse2=2
