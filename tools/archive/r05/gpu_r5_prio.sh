#!/bin/bash
# round 5: wave priority (s_setprio) by the phase of a tile in the headline kernel: diagnostic builds (tools/build_variant.sh priotail
# -DMEMO_PRIO_TAIL=3 | prioscat -DMEMO_PRIO_SCATTER=3 | prioboth "-DMEMO_PRIO_SCATTER=1 -DMEMO_PRIO_TAIL=3") against the product's
# A/B build, one process each, alternating, twice: config 3, the placed six-row view, k = 31 / 21 / 17
TAG=${1:-r5prio}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ab priotail prioscat prioboth; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  echo "== $lib" >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$so timeout 600 python tools/view_sweep_ab.py --ks 31,21,17 --reps 1 --launches 900 --variants 0:6:1 >> $OUT/ab.txt 2>> $OUT/ab.err
done; done
cut -c1-200 $OUT/ab.txt; tail -3 $OUT/ab.err
