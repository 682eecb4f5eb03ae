#!/bin/bash
# membership (config 4) on the dense rows: parity, then sustained runs (one format per process), k = 31 / 21 / 48 / 64
TAG=${1:-md}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or membership or config3_full_size" 2>&1 | tail -5 | tee $OUT/pytest.txt | cut -c1-300
timeout 200 python tests/fuzz_gpu.py --seconds 90 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
for rep in 1 2; do for k in 31 21 48 64; do for rows in only dense; do
  printf "c4 k=%-3s %-6s: " $k $rows >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c4 --k $k --pack $rows --rounds 2000 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
