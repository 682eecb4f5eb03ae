#!/bin/bash
TAG=$1; SECS=${2:-120}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout $((SECS + 120)) python tests/fuzz_dap_gpu.py --seconds $SECS > $OUT/fuzz_dap.txt 2> $OUT/fuzz_dap.err; echo "rc=$?"; tail -3 $OUT/fuzz_dap.txt; grep -v amdgpu.ids $OUT/fuzz_dap.err | tail -5
