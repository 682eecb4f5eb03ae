#!/bin/bash
TAG=${1:-r2x}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "packed_rows or bucket or resident or config5" 2>&1 | tail -3 | cut -c1-200
for rep in 1 2; do for wl in c3 c5; do for k in 101 200 256; do
  U8=""; [ $wl = c3 ] && U8="--u8"
  printf "$wl k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload $wl --k $k --pack only $U8 --rounds 10 "0,0,0" "0,0,0,0,2" "0,0,0,0,3" "2048,8,0,0,2" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2]+j['variant'][4:])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
