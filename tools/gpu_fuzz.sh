#!/bin/bash
TAG=${1:-fz}; SECS=${2:-600}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout $((SECS + 120)) python tests/fuzz_gpu.py --seconds $SECS > $OUT/fuzz.txt 2>&1; tail -3 $OUT/fuzz.txt | cut -c1-600
