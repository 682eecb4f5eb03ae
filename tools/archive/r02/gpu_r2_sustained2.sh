#!/bin/bash
# the level-array choice at k >= 65 re-checked in SUSTAINED runs (one variant per process, 2000 launches back to back)
TAG=${1:-su}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # workload k u8flag variant
  printf "%s k=%-3s %-12s: " $1 $2 $4 >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload $1 --k $2 --pack only $3 --rounds 2000 "$4" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
}
for k in 65 101 128 160 200; do for v in "0,0,0,0,2" "0,0,0,0,3" "0,0,0,0,4"; do run c3 $k --u8 $v; done; done
for k in 101 160; do for v in "0,0,0,0,2" "0,0,0,0,3" "0,0,0,0,4"; do run c5 $k "" $v; done; done
cat $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
