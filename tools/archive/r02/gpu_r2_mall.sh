#!/bin/bash
# the same sweeps on a window whose rows fit the 256 MB Infinity Cache (10^7 positions: 160 / 200 MB of rows), sustained:
# how much of a launch is waiting for HBM?
TAG=${1:-ml}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for rows in only dense; do for L in 10000000 20000000 100000000; do
  printf "c3 k=31 %-6s L=%-10s: " $rows $L >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k 31 --pack $rows --u8 --length $L --rounds 3000 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f   = %.4f ms per 1e8 positions'%(j['ms_median'], j['ms_min'], j['ms_median']*1e8/$L))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
