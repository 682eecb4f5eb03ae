#!/bin/bash
# in-register fold: parity tests, then A/B against the LDS fold passes (build -DMEMO_FOLD_REG=0), 4- and 3-byte rows
TAG=${1:-r2f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "dense or packed_rows or bucket or resident or golden_one_shot or config3 or config5 or randomized or config2 or cli" 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
timeout 100 python tests/fuzz_gpu.py --seconds 60 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
for rep in 1 2 3; do for lib in libmemo_amd_nofoldreg_ab.so libmemo_amd_ab.so; do for k in 21 31 64 101; do
  printf "%-30s k=%-3s: " $lib $k >> $OUT/ab.txt
  V='"0,0,0,0" "0,0,0,2"'; [ $k -gt 64 ] && V='"0,0,0,0"'
  eval MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack both --u8 --rounds 10 $V 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%d B rows %.4f ms (min %.4f)'%(j['row_bytes'], j['ms_median'], j['ms_min']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
