#!/bin/bash
# dense rows as five per 16-byte group: parity, then A/B against the 4-byte rows
TAG=${1:-r2q}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "dense or config3" 2>&1 | tail -5 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
timeout 100 python tests/fuzz_gpu.py --seconds 45 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
for rep in 1 2 3; do for k in 21 31 64; do
  printf "k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack both --u8 --rounds 12 "0,0,0,0" "0,0,0,2" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s B rows %.4f ms (min %.4f)'%(j['row_bytes'], j['ms_median'], j['ms_min']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
