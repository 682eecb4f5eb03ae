#!/bin/bash
# A/B of an environment switch on several workloads: gpu_env_ab.sh tag VAR valA valB -- "wl k pack" ...
TAG=$1; VAR=$2; A=$3; B=$4; shift 5; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for val in $A $B; do for wl in "$@"; do read -r w k pk <<< "$wl"
    printf "%s=%s %s k=%s %s: " $VAR $val $w $k $pk >> $OUT/ab.txt
    PK=""; [ "$pk" != "wide" ] && PK="--pack $pk"
    env $VAR=$val python tools/ab.py --workload $w --k $k $PK --rounds 10 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.3f ms  frac %.3f'%(j['ms_median'], j['frac_of_8TBs']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
