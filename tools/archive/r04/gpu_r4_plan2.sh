#!/bin/bash
# round 4 diagnostic: the plan with every array (scatter 4) on this build against rounds 2-3's clear + fold (-DMEMO_OLD_MIXED build),
# (the -DMEMO_OLD_MIXED diagnostic lived in memo_sweep_cons.hip from commit "Mixed level arrays follow an exact census" until the level plan's own
# clear and fold had caught up -- profiles/r04_large_k.txt, item 3; this script is kept for the record of that run)
# and the repeatability of one process's median (same build, same variant, five processes)
TAG=${1:-r4plan2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # lib workload k pack extra...
  local lib=$1 wl=$2 k=$3 pack=$4; shift 4
  printf "%-12s %s k=%s %s %s: " $lib $wl $k $pack "$*" >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_$lib.so timeout 400 python tools/ab.py --workload $wl --k $k --pack $pack --rounds 1200 "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%s %.4f ms median  min %.4f  frac %.3f'%(j['variant'], j['ms_median'], j['ms_min'], j['frac_of_8TBs']), end='; ')
print()" >> $OUT/ab.txt
}
for rep in 1 2; do for lib in ab oldmixed_ab; do
run $lib c3 256 only --u8 "0,0,0,0,4"
run $lib c3 101 only --u8 "0,0,0,0,4"
done; done
for rep in 1 2 3 4 5; do run ab c3 101 only --u8 "0,0,0"; done
for rep in 1 2 3; do run ab c3 101 only --u8 --row-order 1 "0,0,0"; done
for rep in 1 2 3; do run ab c3 256 only --u8 --row-order 1 "0,0,0,0,4"; done
cat $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
