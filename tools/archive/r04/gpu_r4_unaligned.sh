#!/bin/bash
# round 4: windows whose start is not a multiple of four on the table-driven dense-row kernel (unaligned 4- / 8-byte result stores)
# against the build that sends them to sweep_conservation_halo3_kernel (-DMEMO_TABLE_RASTER_ONLY: rounds 3-4)
TAG=${1:-r4un}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  for qs in 0 1 2; do
    for lib in ab raster_ab; do
      for fl in "--u8" ""; do
        echo -n "c3 k=31 dense qs=$qs $lib ${fl:-u16}: " >> $OUT/ab.txt
        MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_$lib.so python tools/ab.py --workload c3 --k 31 --pack dense $fl --qs $qs --rounds 1200 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  sweep %d' % (j['ms_median'], j['ms_min'], j['last_sweep']))" >> $OUT/ab.txt
      done
    done
  done
done
sort $OUT/ab.txt; tail -2 $OUT/err.txt
