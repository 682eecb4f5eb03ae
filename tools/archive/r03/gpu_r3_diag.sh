#!/bin/bash
# round 3 diagnostic: is the dense-row sweep short of scalar issue?  +50 / +100 do-nothing SALU, +100 VALU per wave (sustained, one build per process)
TAG=${1:-r3d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ab salu50_ab salu100_ab valu100_ab; do
  printf "c3 k=31 dense %-12s: " $lib >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_$lib.so timeout 300 python tools/ab.py --workload c3 --k 31 --pack dense --u8 --rounds 2000 "0,0,0,5" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  frac %.3f'%(j['ms_median'], j['ms_min'], j['frac_of_8TBs']))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
