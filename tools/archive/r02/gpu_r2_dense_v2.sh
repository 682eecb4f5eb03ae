#!/bin/bash
# dense rows re-laid (rows 0-3 used as loaded) against the committed layout: parity, then sustained runs, one build per process
TAG=${1:-d2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or config3_full_size or golden_one_shot" 2>&1 | tail -3 | tee $OUT/pytest.txt | cut -c1-300
timeout 200 python tests/fuzz_gpu.py --seconds 90 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
for rep in 1 2 3; do for lib in libmemo_amd_ab.so libmemo_amd_head_ab.so; do for k in 31 21 64; do
  printf "c3 k=%-3s dense %-26s: " $k $lib >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack dense --u8 --rounds 3000 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
