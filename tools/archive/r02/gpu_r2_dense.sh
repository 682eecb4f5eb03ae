#!/bin/bash
# 3-byte rows: tests, then interleaved A/B of 4-byte vs 3-byte rows on one device, then the bench
TAG=${1:-r2c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "dense or packed_rows or bucket or resident or golden_one_shot or config3 or two_threads or end_before" 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt
for k in 31 21 64; do
  echo "== c3 k=$k u8: 3-byte rows (0,0,0,0) vs 4-byte rows (0,0,0,2)" >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack both --u8 --rounds 12 "0,0,0,0" "0,0,0,2" 2>>$OUT/err.txt >> $OUT/ab.txt
done
cat $OUT/ab.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; cat $OUT/bench_driver.json; grep -v amdgpu.ids $OUT/bench.err | tail -5
timeout 120 python tests/fuzz_gpu.py --seconds 60 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
