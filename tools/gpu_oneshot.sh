#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/oneshot_timing.py > $OUT/oneshot.txt 2>$OUT/err.txt; cat $OUT/oneshot.txt; tail -3 $OUT/err.txt
