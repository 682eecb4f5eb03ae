#!/bin/bash
# 3-byte rows as two planes: tests, A/B against the 4-byte rows; order-in-place A/B (previous build vs this one)
TAG=${1:-r2d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "dense or packed_rows or bucket or golden_one_shot or config3 or multi_device or end_before or split" 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
for k in 31 64; do
  echo "== c3 k=$k u8: library's choice = 4-byte rows (0,0,0,0) vs 3-byte planes (0,0,0,2)" >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack both --u8 --rounds 12 "0,0,0,0" "0,0,0,2" 2>>$OUT/err.txt >> $OUT/ab.txt
done
for rep in 1 2 3; do for lib in libmemo_amd_prev_ab.so libmemo_amd_ab.so; do for k in 31 101; do
  printf "%s k=%s: " $lib $k >> $OUT/ab_libs.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 12 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  frac %.3f'%(j['ms_median'], j['ms_min'], j['frac_of_8TBs']))" >> $OUT/ab_libs.txt
done; done; done
cat $OUT/ab.txt; sort $OUT/ab_libs.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
