#!/bin/bash
# round 3: persistent workgroups (register staging) on the k-class VIEW of the dense rows -- fewer rows per tile, a longer share
# of a tile's life is latency; does a workgroup that prefetches its next tile win there?
TAG=${1:-r3pv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for v in "0,0,0,0" "0,0,0,11" "0,0,0,12"; do for k in 31 21; do
  printf "c3 k=%-3s %-10s: " $k $v >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack dense --u8 --rounds 2000 "$v" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
