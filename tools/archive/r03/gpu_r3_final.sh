#!/bin/bash
# round 3: start-up probe, the widened fuzzer, the full GPU tier once more
TAG=${1:-r3f}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/startup_probe.py 2>&1 | tee $OUT/startup_probe.txt
timeout 500 python tests/fuzz_gpu.py --seconds ${2:-300} > $OUT/fuzz.txt 2>&1; tail -3 $OUT/fuzz.txt | cut -c1-700
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt | cut -c1-300
