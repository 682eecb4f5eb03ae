#!/bin/bash
# the whole GPU tier on the current build: smoke, pytest -m gpu, differential fuzzers, bench (driver's command line)
TAG=${1:-r3z}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt | cut -c1-300
timeout $((${2:-200} + 120)) python tests/fuzz_gpu.py --seconds ${2:-200} > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-400
timeout 200 python tests/fuzz_dap_gpu.py --seconds 40 > $OUT/fuzz_dap.txt 2>&1; tail -2 $OUT/fuzz_dap.txt | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_driver.json
