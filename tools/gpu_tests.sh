#!/bin/bash
TAG=${1:-t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
