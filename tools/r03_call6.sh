source tools/gpu_steps.sh
step r3_t6_stepper 900 python -m pytest tests/test_gpu_stepper.py -x -q -m gpu -k "two_step or strip"
run() { XLBHIP_LIB=$PWD/xlb_amd/lib/$1 python tools/sweep.py --workload $2 --size ${4:-512} --rounds 2 --steps 40 --variant $3 2>/dev/null | grep "^fuse2" | awk -v n="$1 $2 ${4:-512} $3" '{print n, $2, $3}'; }
decomp() {
  for rep in 1 2 3; do
    for w in cavity_halfway periodic cavity_fullway; do
      run r2kernel.so $w fuse2_strips=0
      run libxlbhip.so $w fuse2_strips=0
      run libxlbhip.so $w fuse2_strips=1
      run rowclean.so $w fuse2_strips=0
    done
  done
}
export -f run decomp
step r3_strips_decomp3 1100 bash -c decomp
