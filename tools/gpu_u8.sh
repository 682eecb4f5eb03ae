#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
for u in "" "--u8"; do
  printf "c3 packed %s: " "$u" >> $OUT/u8.txt
  python tools/ab.py --workload c3 --k 31 --pack only $u "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/u8.txt
  printf "c3 wide %s: " "$u" >> $OUT/u8.txt
  python tools/ab.py --workload c3 --k 31 $u "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/u8.txt
done; done
cat $OUT/u8.txt
