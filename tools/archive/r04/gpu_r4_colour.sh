#!/bin/bash
# round 4: the dense views' rows placed inside their groups against LDS bank conflicts (memo_interleave.hip: colour_view_kernel) against
# the order the filter leaves; sustained, one variant per process, alternating
TAG=${1:-r4col}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config3_full_size or dense or view or prepare" > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
for rep in 1 2 3; do
  for k in 31 21 17; do
    for v in "" "--no-colour"; do
      echo -n "c3 k=$k dense ${v:-colour}: " >> $OUT/ab.txt
      python tools/ab.py --workload c3 --k $k --pack dense --u8 --rounds 1500 $v "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  rows_read %d' % (j['ms_median'], j['ms_min'], j['last_rows_read']))" >> $OUT/ab.txt
    done
  done
done
sort $OUT/ab.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2>$OUT/bench.err; python -c "
import json; j=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('view_pass',{}).get('ms'))"
