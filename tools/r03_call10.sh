source tools/gpu_steps.sh
run() { XLBHIP_LIB=$PWD/xlb_amd/lib/$1 python tools/sweep.py --workload $2 --size ${4:-512} --rounds 3 --steps 40 --variant $3 2>/dev/null | grep "^fuse2" | awk -v n="$1 $2 ${4:-512} $3" '{print n, $2, $3}'; }
ab() {
  for rep in 1 2 3; do
    for w in cavity_halfway cavity_fullway; do
      run libxlbhip.so $w fuse2_strips=0
      run libxlbhip.so $w fuse2_strips=1
      run noslack.so $w fuse2_strips=0
      run noslack.so $w fuse2_strips=1
    done
  done
}
export -f run ab
step r3_noslack_ab 1100 bash -c ab
