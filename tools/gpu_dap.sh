#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python tools/dap_timing.py --positions 100000 > $OUT/dap_timing.txt 2>&1
python tools/dap_timing.py --positions 1000000 >> $OUT/dap_timing.txt 2>&1
cat $OUT/dap_timing.txt
