#!/bin/bash
# round 4, first GPU call: what an LDS atomic costs by access pattern (tools/lds_atomic_bench), then same-day sustained baselines
# of the lines VERDICT r03 names (k >= 65, membership at k = 101) and of the headline, one variant per process
TAG=${1:-r4base}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 tools/lds_atomic_bench > $OUT/lds_atomic_bench.txt 2>&1
run() {  # workload k pack extra...
  local wl=$1 k=$2 pack=$3; shift 3
  printf "%s k=%s %s %s: " $wl $k $pack "$*" >> $OUT/ab.txt
  timeout 400 python tools/ab.py --workload $wl --k $k --pack $pack --rounds 1500 "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%s %.4f ms median  min %.4f  frac %.3f'%(j['variant'], j['ms_median'], j['ms_min'], j['frac_of_8TBs']), end='; ')
print()" >> $OUT/ab.txt
}
run c3 101 only --u8 "0,0,0"
run c3 256 only --u8 "0,0,0"
run c5 101 only "0,0,0"
run c4 101 only "0,0,0"
run c3 31 dense --u8 "0,0,0"
run c3 21 dense --u8 "0,0,0"
run c4 31 only "0,0,0"
run c5 31 only "0,0,0"
cat $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
