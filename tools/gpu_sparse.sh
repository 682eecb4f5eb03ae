#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $*" >> $OUT/sparse.txt; python tools/ab.py "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['variant'], '%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/sparse.txt; }
run --workload sparse --k 31 "0,0,0" "1024,4,0" "4096,4,0"
run --workload c2 --k 31 "0,0,0" "512,1,0" "512,4,0" "256,1,0" "256,4,0"
run --workload c2 --k 31 --pack only "0,0,0" "512,1,0" "512,4,0" "256,1,0"
run --workload c3 --k 31 "0,0,0" "4096,4,0"
cat $OUT/sparse.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
