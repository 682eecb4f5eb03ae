#!/bin/bash
# mixed-level kernel: the long intervals as one branch-free block, against the branchy form: parity, then sustained
TAG=${1:-cy}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_arrays or config3_full_size or config5 or packed or resident" 2>&1 | tail -3 | cut -c1-300
timeout 200 python tests/fuzz_gpu.py --seconds 120 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
run() {
  for lib in libmemo_amd_ab.so libmemo_amd_nocmpx_ab.so; do
    printf "%s k=%-3s %-26s: " $1 $2 $lib >> $OUT/ab.txt
    MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload $1 --k $2 --pack only $3 --rounds 2000 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
  done
}
for rep in 1 2; do run c3 101 --u8; run c3 80 --u8; run c3 256 --u8; run c5 101 ""; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
