source tools/gpu_steps.sh
# where does the strip-buffer build lose?  same box, separate processes, two repetitions each
run() { XLBHIP_LIB=$PWD/xlb_amd/lib/$1 python tools/sweep.py --workload $2 --size 512 --rounds 2 --steps 40 --variant $3 2>/dev/null | grep "^fuse2" | awk -v n="$1 $2 $3" '{print n, $2, $3}'; }
decomp() {
  for rep in 1 2; do
    for w in cavity_halfway periodic; do
      run libxlbhip.so $w fuse2_strips=0
      run libxlbhip.so $w fuse2_strips=1
      run rowmap.so $w fuse2_strips=0
      run strips_nowrite.so $w fuse2_strips=1
      run strips_noread.so $w fuse2_strips=1
    done
  done
}
export -f run decomp
step r3_strips_decomp 1100 bash -c decomp
