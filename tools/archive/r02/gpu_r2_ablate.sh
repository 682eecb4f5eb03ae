#!/bin/bash
# what each phase of the unclipped conservation sweep costs: diagnostic builds with phases removed (results
# are wrong in them), one process per build, three rounds, config 3 packed rows, k = 31 and 101
TAG=${1:-r2e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
LIBS="${2:-libmemo_amd_ab.so libmemo_amd_a1_ab.so libmemo_amd_a2_ab.so libmemo_amd_a3_ab.so libmemo_amd_a4_ab.so libmemo_amd_a8_ab.so libmemo_amd_a16_ab.so libmemo_amd_a31_ab.so}"
for rep in 1 2 3; do for lib in $LIBS; do for k in 31 101; do
  printf "%-28s k=%-3s: " $lib $k >> $OUT/ablate.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 10 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ablate.txt
done; done; done
sort $OUT/ablate.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
