#!/bin/bash
# A/B of AB-library builds (tools/build_variant.sh): gpu_ab.sh tag "wl k pack" ... -- libmemo_amd_X_ab.so ...
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
WLS=(); while [ "$1" != "--" ]; do WLS+=("$1"); shift; done; shift
LIBS=("$@")
for rep in 1 2; do for lib in "${LIBS[@]}"; do for wl in "${WLS[@]}"; do read -r w k pk <<< "$wl"
    printf "%s %s k=%s %s: " $lib $w $k $pk >> $OUT/ab.txt
    PK=""; [ "$pk" != "wide" ] && PK="--pack $pk"
    MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib python tools/ab.py --workload $w --k $k $PK --rounds 10 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
