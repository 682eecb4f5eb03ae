#!/bin/bash
# radix-4 kernel on an instruction diet (scatter 21 -> 17 VALU per row, register fold through DPP operands) against
# the previous one: parity (fuzz covers scatter = 3 at every k), then one process per build, alternating
TAG=${1:-r4c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tests/fuzz_gpu.py --seconds 120 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
timeout 900 python -m pytest tests -x -q -m gpu -k "k101 or radix or r4 or config3 or config5 or packed or golden_one_shot" 2>&1 | tail -3 | tee $OUT/pytest.txt | cut -c1-300
LIBS="libmemo_amd_ab.so libmemo_amd_r4prev_ab.so"
for rep in 1 2 3; do for lib in $LIBS; do for k in 101 200 256 65; do
  printf "c3 k=%-3s %-28s: " $k $lib >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 10 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
for rep in 1 2; do for lib in $LIBS; do
  printf "c5 k=101 r4 %-25s: " $lib >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c5 --k 101 --pack only --rounds 8 "0,0,0,0,3" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
