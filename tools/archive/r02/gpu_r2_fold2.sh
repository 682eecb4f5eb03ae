#!/bin/bash
# DPP-fused register fold against the previous one: parity first, then one process per build, alternating
TAG=${1:-fd}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or config3 or packed or conservation or fuzz or dense" 2>&1 | tail -4 | tee $OUT/pytest.txt | cut -c1-300
timeout 200 python tests/fuzz_gpu.py --seconds 90 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
LIBS="libmemo_amd_ab.so libmemo_amd_olddfold_ab.so"
for rep in 1 2 3; do for lib in $LIBS; do for k in 31 21 64; do
  printf "%-32s k=%-3s: " $lib $k >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 10 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
for rep in 1 2; do for lib in $LIBS; do
  printf "%-32s dense k=31: " $lib >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k 31 --pack dense --u8 --rounds 10 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
