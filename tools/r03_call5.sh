source tools/gpu_steps.sh
step r3_t5_stepper 900 python -m pytest tests/test_gpu_stepper.py tests/test_gpu_fullsize.py -x -q -m gpu -k "two_step or strip or tiling or c_oracle"
run() { XLBHIP_LIB=$PWD/xlb_amd/lib/$1 python tools/sweep.py --workload $2 --size ${4:-512} --rounds 2 --steps 40 --variant $3 2>/dev/null | grep "^fuse2" | awk -v n="$1 $2 ${4:-512} $3" '{print n, $2, $3}'; }
decomp() {
  for rep in 1 2 3; do
    for w in cavity_halfway periodic cavity_fullway; do
      run r2kernel.so $w fuse2_strips=0
      run libxlbhip.so $w fuse2_strips=0
      run libxlbhip.so $w fuse2_strips=1
      run strips_nowrite.so $w fuse2_strips=1
    done
  done
  run r2kernel.so periodic fuse2_strips=0 256
  run libxlbhip.so periodic fuse2_strips=0 256
  run libxlbhip.so periodic fuse2_strips=1 256
}
export -f run decomp
step r3_strips_decomp2 1100 bash -c decomp
SWEEP_ARGS="--size 384 --lattice D3Q27" VARIANT="fuse2=2" step r3_d3q27_rowmap_ab 600 bash tools/ab_libs.sh periodic 2 xlb_amd/lib/r2kernel.so xlb_amd/lib/libxlbhip.so
step r3_bigcopy 600 python tools/check_big_copy.py 4096
