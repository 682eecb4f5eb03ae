#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
run() { echo "== $*" >> $OUT/memb.txt; python tools/ab.py "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['variant'], '%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/memb.txt; }
run --workload c4 --k 31 --pack only "0,0,0" "256,4,2" "1024,4,3" "2048,4,3" "4096,4,3" "512,4,3" "1024,1,3"
run --workload c4 --k 31 "0,0,0" "512,4,2" "1024,4,3" "2048,4,3" "4096,4,3"
run --workload c4 --k 101 --pack only "256,4,2" "2048,4,3" "1024,4,3"
run --workload c4 --k 21 --pack only "256,4,2" "2048,4,3"
cat $OUT/memb.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
