#!/bin/bash
# array size x waves for the doubling arrays at k = 31 / 21 after the fold's instruction diet (interleaved, one process per k)
TAG=${1:-sh}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do for k in 31 21; do
  printf "c3 k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 10 "0,0,0" "1024,8,0" "1280,4,0" "1536,4,0" "1536,8,0" "2048,8,0" "768,4,0" "2048,4,0" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
