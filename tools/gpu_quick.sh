#!/bin/bash
TAG=${1:-q}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -6 | tee $OUT/pytest.txt | cut -c1-300
timeout 100 python tests/fuzz_gpu.py --seconds 45 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
