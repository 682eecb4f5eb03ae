#!/bin/bash
# round 4: the table-driven kernel's row blocks without the "this row writes" test on a view of exactly the writing rows (AW), against
# the blocks with it (memo_debug_set_tuning row_source 13); sustained, one variant per process
TAG=${1:-r4aw}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or views or config3_full or 256_to_511 or places" 2>&1 | tail -3 | tee $OUT/pytest.txt
for rep in 1 2 3; do
  for spec in "c3 31 --u8" "c3 21 --u8" "c3 17 --u8" "c5 31 "; do read -r w k fl <<< "$spec"
    for v in "0,0,0" "0,0,0,13"; do
      echo -n "$w k=$k dense $v: " >> $OUT/ab.txt
      python tools/ab.py --workload $w --k $k --pack dense $fl --rounds 1500 "$v" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  rows_read %d' % (j['ms_median'], j['ms_min'], j['last_rows_read']))" >> $OUT/ab.txt
    done
  done
done
sort $OUT/ab.txt; tail -2 $OUT/err.txt
