#!/bin/bash
# 4-byte rows against dense rows in SUSTAINED runs (one format per process, thousands of launches back to back: the
# device settles at its power cap), alternating processes, several k
TAG=${1:-su}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for k in 31 21 48 64; do for rows in only dense; do
  printf "c3 k=%-3s %-6s: " $k $rows >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack $rows --u8 --rounds 3000 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
