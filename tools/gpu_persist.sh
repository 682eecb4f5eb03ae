#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest.txt; tail -2 $OUT/pytest.txt
run() { echo "== $*" >> $OUT/persist.txt; python tools/ab.py "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['variant'], '%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/persist.txt; }
run --workload c3 --k 31 --pack only "0,0,0,1" "0,0,0,2" "256,1,0,2" "512,4,0,2" "2048,4,0,2"
run --workload c3 --k 101 --pack only "0,0,0,1" "0,0,0,2"
run --workload c5 --k 31 --pack only "0,0,0,1" "0,0,0,2"
run --workload c4 --k 31 --pack only "0,0,0,1" "0,0,0,2"
run --workload c3 --k 31 "0,0,0,1" "0,0,0,2"
run --workload c4 --k 31 "0,0,0,1" "0,0,0,2"
run --workload c2 --k 31 --pack only "0,0,0,1" "0,0,0,2"
cat $OUT/persist.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
