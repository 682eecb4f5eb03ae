#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $*" >> $OUT/memb.txt; python tools/ab.py "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['variant'], '%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/memb.txt; }
run --workload c4 --k 31 --pack only "256,4,3" "512,4,3" "1024,4,3" "2048,4,3" "4096,4,3" "512,1,3" "1024,1,3" "256,4,2"
run --workload c4 --k 101 --pack only "512,4,3" "1024,4,3" "2048,4,3" "4096,4,3"
run --workload c4 --k 31 "512,4,2" "1024,4,3" "2048,4,3" "4096,4,3"
cat $OUT/memb.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
