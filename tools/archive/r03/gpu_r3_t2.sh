#!/bin/bash
# round 3: the table-driven kernel after the row blocks were merged (sustained), then the index from sequences
TAG=${1:-r3t2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/p3_check.py 2>&1 | tail -5
for rep in 1 2; do for v in "0,0,0,5" "0,0,0,8"; do for k in 31 64; do
  printf "c3 k=%-3s %-10s: " $k $v >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack dense --u8 --rounds 2000 "$v" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  frac %.3f'%(j['ms_median'], j['ms_min'], j['frac_of_8TBs']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
bash tools/gpu_r3_real.sh ${TAG}_real
