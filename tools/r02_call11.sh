#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/c11
S="python tools/sweep.py --size 512 --rounds 3 --steps 20"
$S --workload periodic --variant exact_math=1 --variant exact_math=0 2>/dev/null | tee gpurun_out/c11/p.txt
$S --workload cavity_halfway --variant exact_math=1 --variant exact_math=0 2>/dev/null | tee gpurun_out/c11/h.txt
$S --workload cavity_fullway --variant exact_math=1 --variant exact_math=0 2>/dev/null | tee gpurun_out/c11/f.txt
python tools/sweep.py --size 384 --lattice D3Q27 --rounds 3 --steps 20 --workload periodic --variant exact_math=1 --variant exact_math=0 2>/dev/null | tee gpurun_out/c11/d27.txt
