#!/bin/bash
# the library's own choice of level arrays ("0,0,0") against each forced scheme; mixed-array shapes at six levels
TAG=${1:-mx}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_arrays or sidecar or builder or golden_one_shot" 2>&1 | tail -5 | tee $OUT/pytest.txt | cut -c1-300
timeout 300 python tests/fuzz_gpu.py --seconds ${2:-90} > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-400
for rep in 1 2; do for k in 65 101 128 129 160 200 256; do
  printf "c3 k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 8 "0,0,0" "0,0,0,0,2" "0,0,0,0,3" "0,0,0,0,4" "1792,8,0,0,4" "2048,8,0,0,4" "1408,8,0,0,4" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2]+j['variant'][4:])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
for rep in 1 2; do for k in 65 101 128 200; do
  printf "c5 k=%-3s: " $k >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c5 --k $k --pack only --rounds 6 "0,0,0" "0,0,0,0,2" "0,0,0,0,3" "0,0,0,0,4" "2048,8,0,0,4" 2>>$OUT/err.txt | python -c "
import json,sys
print(' | '.join('%s %.4f'%(','.join(map(str,j['variant'][:2]+j['variant'][4:])), j['ms_median']) for j in map(json.loads, sys.stdin)))" >> $OUT/ab.txt
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
