#!/bin/bash
# round 4: order of the 4-byte rows inside a bucket (memo_interleave.hip): the GPU tier on the new default, then sustained A/B of
# the orders (1 start order, 2 chunks of four dealt over the starts, 3 + by overlap mod 32), one process per order
TAG=${1:-r4order}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "$2" != "notests" ]; then timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt; fi
run() {  # workload k pack order extra...
  local wl=$1 k=$2 pack=$3 ord=$4; shift 4
  printf "%s k=%s %s order %s %s: " $wl $k $pack $ord "$*" >> $OUT/ab.txt
  timeout 400 python tools/ab.py --workload $wl --k $k --pack $pack --row-order $ord --rounds 1200 "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%s %.4f ms median  min %.4f  frac %.3f'%(j['variant'], j['ms_median'], j['ms_min'], j['frac_of_8TBs']), end='; ')
print()" >> $OUT/ab.txt
}
for rep in 1 2; do for ord in 1 2 3; do
run c5 101 only $ord "0,0,0"
run c3 101 only $ord --u8 "0,0,0"
run c3 256 only $ord --u8 "0,0,0"
run c5 31 only $ord "0,0,0"
run c4 101 only $ord "0,0,0"
run c4 31 only $ord "0,0,0"
run c3 31 only $ord --u8 "0,0,0,3"
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
