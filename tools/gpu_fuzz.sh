#!/bin/bash
TAG=$1; SECS=${2:-300}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout $((SECS + 120)) python tests/fuzz_gpu.py --seconds $SECS > $OUT/fuzz.txt 2> $OUT/fuzz.err; echo "rc=$?"; tail -3 $OUT/fuzz.txt; grep -v amdgpu.ids $OUT/fuzz.err | tail -5
