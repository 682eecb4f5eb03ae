#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() { lib=$1; shift; echo "== $lib $*" >> $OUT/shapes.txt; MEMO_AMD_LIB=$PWD/memo_amd/$lib python tools/ab.py "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['variant'], '%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/shapes.txt; }
for lib in libmemo_amd.so libmemo_amd_u2.so libmemo_amd_u8.so; do
  run $lib --workload c3 --k 31 --pack only "256,1,0" "512,1,0" "512,4,0" "1024,4,0" "2048,4,0"
  run $lib --workload c3 --k 101 --pack only "256,1,0" "256,4,0" "512,4,0" "1024,4,0"
  run $lib --workload c4 --k 31 --pack only "256,4,2" "512,4,2" "256,1,2"
  run $lib --workload c5 --k 31 --pack only "512,4,0" "1024,4,0" "2048,4,0"
done
cat $OUT/shapes.txt
