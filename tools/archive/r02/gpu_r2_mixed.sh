#!/bin/bash
# mixed level arrays (scatter 4) against radix-4 (3) and doubling (2): parity by fuzz, then interleaved A/B per k
TAG=${1:-mx}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tests/fuzz_gpu.py --seconds ${2:-120} > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-400
for rep in 1 2; do for k in 65 80 101 128 129 200 256; do
  printf "c3 k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 8 "0,0,0,0,3" "0,0,0,0,2" "0,0,0,0,4" "2560,8,0,0,4" "1536,8,0,0,4" "1536,4,0,0,4" "1280,4,0,0,4" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2]+j['variant'][4:])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
for rep in 1 2; do for k in 65 101 200; do
  printf "c5 k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c5 --k $k --pack only --rounds 6 "0,0,0,0,3" "0,0,0,0,2" "0,0,0,0,4" "2560,8,0,0,4" "1536,8,0,0,4" "1280,4,0,0,4" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2]+j['variant'][4:])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
