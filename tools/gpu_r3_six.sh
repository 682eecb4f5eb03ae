#!/bin/bash
# round 3: the bucket-grouped form of the k-class views (six 18-bit rows per 16 bytes; row_source 13) against the views as they
# are (five rows per 16 bytes; row_source 0): parity gate first, then sustained A/B on BASELINE config 3
TAG=${1:-r3six}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 800 python tools/six_check.py 2>&1 | tail -1
for rep in 1 2; do for v in "0,0,0,0" "0,0,0,13"; do for k in 31 21 17 9; do
  printf "c3 k=%-3s %-10s: " $k $v >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack dense --u8 --rounds 1500 "$v" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
