#!/bin/bash
# A/B of several builds of the library on one box: every build is timed REPS times in alternation (separate processes).
#   tools/ab_libs.sh "periodic cavity_halfway" 2 lib1.so lib2.so ...
#   SWEEP_ARGS="--size 384 --lattice D3Q27 --collision KBC" VARIANT="fuse2=0" tools/ab_libs.sh periodic 2 a.so b.so
WL=$1; REPS=$2; shift 2
for r in $(seq $REPS); do
  for L in "$@"; do
    for w in $WL; do
      V=${VARIANT:-fuse2=2}
      t=$(XLBHIP_LIB=$PWD/$L python tools/sweep.py --workload $w ${SWEEP_ARGS:---size 512} --rounds 2 --steps 40 --variant $V 2>/dev/null | grep "^$V" | awk '{print $2}') || exit 1
      echo "$(basename $L) $w rep$r $t"
    done
  done
done
