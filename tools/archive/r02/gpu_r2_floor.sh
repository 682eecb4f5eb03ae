#!/bin/bash
# memory floors: rows loaded and dropped (build -DMEMO_ABLATE=16; =28 also without clear and fold), 4-byte rows vs dense rows
TAG=${1:-r2r}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do for lib in libmemo_amd_ab.so libmemo_amd_a16_ab.so libmemo_amd_a28_ab.so; do for k in 31; do
  printf "%-26s k=%-3s: " $lib $k >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack both --u8 --rounds 12 "0,0,0,0" "0,0,0,2" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s B rows %.4f ms (min %.4f)'%(j['row_bytes'], j['ms_median'], j['ms_min']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
