#!/bin/bash
# gpu_libs_ab.sh TAG "wl k" variant "ab.py flags" lib... : one ab.py process per library build (not interleaved)
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; read -r w k <<< "$2"; VAR=$3; FLAGS=$4; shift 4
for rep in 1 2; do for lib in "$@"; do
  printf "%-12s " $lib >> $OUT/libs.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_$lib.so python tools/ab.py --workload $w --k $k --pack only $FLAGS --rounds 12 "$VAR" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms (min %.4f)'%(j['ms_median'], j['ms_min']))" >> $OUT/libs.txt
done; done
cat $OUT/libs.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
