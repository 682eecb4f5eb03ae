#!/bin/bash
# dense rows, config 3, k = 31: tile shapes in SUSTAINED runs (one variant per process, 3000 launches back to back)
TAG=${1:-su}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for v in "0,0,0" "768,4,0" "896,4,0" "512,4,0" "1024,8,0" "640,4,0"; do
  printf "c3 k=31 dense %-10s: " $v >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k 31 --pack dense --u8 --rounds 3000 "$v" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
