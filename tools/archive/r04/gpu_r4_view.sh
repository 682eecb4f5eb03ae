#!/bin/bash
# round 4, VERDICT r03 item 4: the view regime's per-tile LDS traffic.  The level arrays cleared by ds_write_addtid_b32 (the
# product) against one ds_write_b128 per lane and level (-DMEMO_CLEAR_B128 build: round 3), sustained, on the k-class views
TAG=${1:-r4view}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # lib k extra...
  local lib=$1 k=$2; shift 2
  printf "%-8s c3 k=%s dense %s: " $lib $k "$*" >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_$lib.so timeout 400 python tools/ab.py --workload c3 --k $k --pack dense --u8 --rounds 1500 "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%s %.4f ms median  min %.4f  rows_read %d'%(j['variant'], j['ms_median'], j['ms_min'], j['last_rows_read']), end='; ')
print()" >> $OUT/ab.txt
}
for rep in 1 2 3; do for lib in ab b128_ab; do for k in 31 21 17 64; do run $lib $k "0,0,0"; done; run $lib 31 "0,0,0,9"; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
