#!/bin/bash
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu "$@" 2>&1 | tail -15 | tee $OUT/pytest.txt
