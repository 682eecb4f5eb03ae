#!/bin/bash
# doubling arrays (register fold up to 7 levels) against radix-4 arrays from k = 65 up, one process per k, interleaved
TAG=${1:-r4d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for k in 65 80 101 128 129 200; do
  printf "c3 k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 8 "0,0,0,0,3" "0,0,0,0,2" "1280,4,0,0,2" "1536,8,0,0,2" "2048,8,0,0,2" "2560,8,0,0,3" "2048,8,0,0,3" "1280,4,0,0,3" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2]+j['variant'][4:])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
