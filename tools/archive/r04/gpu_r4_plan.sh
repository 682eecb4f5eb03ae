#!/bin/bash
# round 4: the level plan of the mixed arrays (only populated levels exist): parity subset, then sustained timings, library's
# plan (variant 0,0,0) against all the arrays (scatter 4: rounds 2-3)
TAG=${1:-r4plan}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_arrays or full_size or config5 or row_order or 120" 2>&1 | tail -8 | tee $OUT/pytest.txt
run() {  # workload k pack extra...
  local wl=$1 k=$2 pack=$3; shift 3
  printf "%s k=%s %s %s: " $wl $k $pack "$*" >> $OUT/ab.txt
  timeout 400 python tools/ab.py --workload $wl --k $k --pack $pack --rounds 1200 "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%s %.4f ms median  min %.4f  frac %.3f'%(j['variant'], j['ms_median'], j['ms_min'], j['frac_of_8TBs']), end='; ')
print()" >> $OUT/ab.txt
}
for rep in 1 2; do
for v in "0,0,0" "0,0,0,0,4"; do
run c3 101 only --u8 $v
run c3 256 only --u8 $v
run c3 128 only --u8 $v
run c3 80 only --u8 $v
run c5 101 only $v
run c5 200 only $v
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
