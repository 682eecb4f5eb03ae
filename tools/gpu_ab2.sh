#!/bin/bash
# A/B of two library builds on the packed workloads.  usage: gpu_ab2.sh tag libA libB ...
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for lib in "$@"; do
  for wl in "c3 31" "c3 101" "c5 31" "c4 31"; do set -- $wl
    echo "== $lib $1 k=$2" >> $OUT/ab.txt
    MEMO_AMD_LIB=$PWD/memo_amd/$lib python tools/ab.py --workload $1 --k $2 --pack only --rounds 10 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['variant'], '%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
