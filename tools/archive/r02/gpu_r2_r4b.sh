#!/bin/bash
# radix-4 levels with the hand-written scatter: parity (k >= 65 is its default), A/B over k and array sizes
TAG=${1:-r2i}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "dense or packed_rows or bucket or resident or config5 or randomized" 2>&1 | tail -6 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
timeout 100 python tests/fuzz_gpu.py --seconds 45 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-400
for rep in 1 2; do for k in 21 31 64 101 256; do
  printf "k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 10 "0,0,0,0,2" "1024,4,0,0,3" "1280,4,0,0,3" "1728,4,0,0,3" "2560,4,0,0,3" "2560,8,0,0,3" "3392,8,0,0,3" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2]+j['variant'][4:])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
