#!/bin/bash
# round 3: phase ablation of the table-driven dense-row sweep (diagnostic builds -DMEMO_T_ABLATE=bits: 1 rows dropped, 2 no clear,
# 4 no fold, 8 no store, 16 no row loads), on the k-class view (row_source 0) and on all the dense rows (9); sustained
TAG=${1:-r3tab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ab tab1_ab tab2_ab tab4_ab tab8_ab tab16_ab tab17_ab tab31_ab; do for v in "0,0,0,0" "0,0,0,9"; do
  printf "c3 k=31 %-9s %-8s: " $lib $v >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_$lib.so timeout 300 python tools/ab.py --workload c3 --k 31 --pack dense --u8 --rounds 1500 "$v" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
