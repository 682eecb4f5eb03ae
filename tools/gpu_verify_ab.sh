#!/bin/bash
# parity tests on the current build, then A/B against libmemo_amd_base.so
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.txt; tail -2 $OUT/pytest_gpu.txt
bash tools/gpu_ab.sh $TAG "c3 31 only" "c3 101 only" "c5 31 only" "c4 31 only" "c3 31 wide" "c3 101 wide" "c4 31 wide" -- libmemo_amd_base.so libmemo_amd.so
