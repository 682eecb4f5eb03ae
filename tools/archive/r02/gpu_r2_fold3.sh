#!/bin/bash
# how many doubling levels to fold in registers (DPP fold): 5 (default), 6, 7; one process per build, alternating
TAG=${1:-fl}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
LIBS="libmemo_amd_ab.so libmemo_amd_fl6_ab.so libmemo_amd_fl7_ab.so"
for lib in libmemo_amd_fl7_ab.so; do   # parity of the 6- and 7-level steps before timing them
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib MEMO_AMD_LIB=$PWD/memo_amd/$lib timeout 200 python tests/fuzz_gpu.py --seconds 60 > $OUT/fuzz_$lib.txt 2>&1; tail -1 $OUT/fuzz_$lib.txt | cut -c1-300
done
for rep in 1 2; do for lib in $LIBS; do
 for spec in "c3 64 --u8 0,0,0" "c3 101 --u8 0,0,0,0,2" "c3 128 --u8 0,0,0,0,2" "c5 64 - 0,0,0" "c5 101 - 0,0,0" ; do
  set -- $spec; U8=$3; [ "$U8" = "-" ] && U8=""
  printf "%-5s k=%-3s %-28s: " $1 $2 $lib >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload $1 --k $2 --pack only $U8 --rounds 8 "$4" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
 done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
