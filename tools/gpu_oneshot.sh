#!/bin/bash
TAG=${1:-os}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/oneshot.txt
timeout 900 python tools/oneshot_timing.py >> $OUT/oneshot.txt 2>&1; grep -v amdgpu.ids $OUT/oneshot.txt
