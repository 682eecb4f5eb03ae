#!/bin/bash
# round 2, first GPU pass: the whole -m gpu suite, the driver's bench command line, the default bench
TAG=${1:-r2a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$?"
cat $OUT/bench_driver.json; grep -v amdgpu.ids $OUT/bench_driver.err | tail -5
timeout 600 python bench.py --cpu-sample 0 > $OUT/bench_default.json 2>> $OUT/bench_driver.err; cat $OUT/bench_default.json
