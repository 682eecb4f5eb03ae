#!/bin/bash
# quick look at the library's own choices on the standard workloads: gpu_quick_ab.sh tag
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for wl in "c3 31 only" "c3 31 wide" "c2 31 only" "c2 31 wide" "c5 31 only" "c4 31 only" "c3 101 wide"; do read -r w k pk <<< "$wl"
  PK=""; [ "$pk" != "wide" ] && PK="--pack $pk"
  printf "%s k=%s %s: " $w $k $pk >> $OUT/quick.txt
  python tools/ab.py --workload $w --k $k $PK --rounds 10 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/quick.txt
done; cat $OUT/quick.txt
