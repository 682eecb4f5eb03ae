#!/bin/bash
# round 4: windows off the 4-position raster on the 4-byte rows' unclipped kernels (halo: register fold instead of the LDS passes; r4 /
# mixed: one store of four instead of four scalar ones) against a build from before (libmemo_amd_raster_ab.so)
TAG=${1:-r4un3}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
run() {  # workload k flags
  for qs in 0 1; do
    for lib in ab raster_ab; do
      echo -n "$1 k=$2 only $3 qs=$qs $lib: " >> $OUT/ab.txt
      MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_$lib.so python tools/ab.py --workload $1 --k $2 --pack only $3 --qs $qs --rounds 1000 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  sweep %d' % (j['ms_median'], j['ms_min'], j['last_sweep']))" >> $OUT/ab.txt
    done
  done
}
run c3 31 --u8; run c3 31 ""; run c3 101 --u8; run c3 256 --u8; run c5 101 ""; run c5 31 ""
sort $OUT/ab.txt; tail -2 $OUT/err.txt
