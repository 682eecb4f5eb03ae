#!/bin/bash
# the tile-locate diagnostic again, in SUSTAINED runs and on the dense rows (which sit at the tile pipeline's own cost)
TAG=${1:-lo}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do for lib in libmemo_amd_ab.so libmemo_amd_synthlocate_ab.so; do for rows in dense only; do
  printf "c3 k=31 %-6s %-32s: " $rows $lib >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k 31 --pack $rows --u8 --rounds 3000 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
