#!/bin/bash
# round 4 experiment: dense k-class views as groups of SIX rows that carry their bucket (memo_debug_six_views) against the five-row groups
TAG=${1:-r4six}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k six_row 2>&1 | tail -3 | tee $OUT/check.txt
for rep in 1 2; do
  for k in 31 21 17; do
    for v in "" "--six"; do
      echo -n "c3 k=$k dense ${v:-five}: " >> $OUT/ab.txt
      python tools/ab.py --workload c3 --k $k --pack dense --u8 --rounds 1500 $v "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  rows_read %d' % (j['ms_median'], j['ms_min'], j['last_rows_read']))" >> $OUT/ab.txt
    done
  done
done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
