#!/bin/bash
# dense rows with the table-driven scatter: parity, then A/B (4-byte rows | dense rows) for the LUT build and the computed build
TAG=${1:-r2s}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "dense or config3" 2>&1 | tail -5 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
timeout 100 python tests/fuzz_gpu.py --seconds 45 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
for rep in 1 2 3; do for lib in libmemo_amd_ab.so libmemo_amd_nolut_ab.so; do for k in 21 31 64; do
  printf "%-26s k=%-3s: " $lib $k >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack both --u8 --rounds 12 "0,0,0,0" "0,0,0,2" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s B rows %.4f ms (min %.4f)'%(j['row_bytes'], j['ms_median'], j['ms_min']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
