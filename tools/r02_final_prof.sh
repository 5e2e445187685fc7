#!/bin/bash
# round-2 evidence: profiles (kernel trace + PMC passes) of the headline config and of configs[1] / configs[4], the DESIGN table
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash tools/profile.sh r02_cavity_halfway_512 > gpurun_out/prof_a.log 2>&1; echo "cavity rc=$?"
bash tools/profile.sh r02_periodic_256 --workload periodic --size 256 > gpurun_out/prof_b.log 2>&1; echo "p256 rc=$?"
bash tools/profile.sh r02_d3q27_kbc_384_fp64fp32_fast --workload periodic --size 384 --lattice D3Q27 --collision KBC --policy FP64FP32 --omega 1.9 --steps 100 > gpurun_out/prof_c.log 2>&1; echo "kbc64 rc=$?"
bash tools/profile.sh r02_d3q27_bgk_384_two_step --workload periodic --size 384 --lattice D3Q27 --steps 100 > gpurun_out/prof_d.log 2>&1; echo "d3q27 rc=$?"
bash tools/profile.sh r02_cavity_fullway_512 --workload cavity_fullway > gpurun_out/prof_e.log 2>&1; echo "fullway rc=$?"
bash tools/profile.sh r02_periodic_512 --workload periodic > gpurun_out/prof_f.log 2>&1; echo "p512 rc=$?"
