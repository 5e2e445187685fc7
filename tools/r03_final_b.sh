#!/bin/bash
# round-3 evidence, part B (one gpurun call): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the headline configurations
source tools/gpu_steps.sh
step r3_prof_cavity_halfway 500 bash tools/profile.sh r03_cavity_halfway_512
step r3_prof_cavity_fullway 500 bash tools/profile.sh r03_cavity_fullway_512 --workload cavity_fullway
step r3_prof_periodic_512 500 bash tools/profile.sh r03_periodic_512 --workload periodic
step r3_prof_periodic_256 400 bash tools/profile.sh r03_periodic_256 --workload periodic --size 256
step r3_prof_kbc 500 bash tools/profile.sh r03_d3q27_kbc_384_fp64fp32 --workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9 --steps 100
step r3_prof_d3q27 500 bash tools/profile.sh r03_d3q27_bgk_384_two_step --workload periodic --size 384 --lattice D3Q27 --steps 100
