#!/bin/bash
# full-size parity of the k = 101 / 128 kernels, the six-array threshold on a dense index, a long fuzz
TAG=${1:-lt}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config3_full_size" 2>&1 | tail -4 | tee $OUT/pytest.txt | cut -c1-300
for rep in 1 2; do for k in 140 160 180; do
  printf "c5 k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c5 --k $k --pack only --rounds 6 "0,0,0" "0,0,0,0,2" "0,0,0,0,3" "0,0,0,0,4" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2]+j['variant'][4:])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
timeout $((${2:-600} + 120)) python tests/fuzz_gpu.py --seconds ${2:-600} > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
