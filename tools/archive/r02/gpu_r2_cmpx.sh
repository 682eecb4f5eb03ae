#!/bin/bash
# rows as one branch-free block (v_cmpx writes EXEC) against the branchy form (-DMEMO_ROW_CMPX=0): parity, then sustained
TAG=${1:-cx}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not sidecar and not cli" 2>&1 | tail -3 | cut -c1-300
timeout 200 python tests/fuzz_gpu.py --seconds 120 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
run() {  # workload k pack u8
  for lib in libmemo_amd_ab.so libmemo_amd_nocmpx_ab.so; do
    printf "%s k=%-3s %-6s %-26s: " $1 $2 $3 $lib >> $OUT/ab.txt
    MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload $1 --k $2 --pack $3 $4 --rounds 2500 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
  done
}
for rep in 1 2; do run c3 31 dense --u8; run c3 31 only --u8; run c3 64 only --u8; run c5 31 only ""; run c5 21 only ""; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
