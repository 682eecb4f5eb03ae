#!/bin/bash
TAG=${1:-cli}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo skip-tests
timeout 900 python tools/cli_timing.py --num-docs 100 --pivot 20000000 --out /tmp/cli_t > $OUT/cli_timing.txt 2>&1; cat $OUT/cli_timing.txt
MEMO_DECODE_THREADS=1 timeout 900 python tools/cli_timing.py --num-docs 100 --pivot 20000000 --out /tmp/cli_t > $OUT/cli_timing_1decoder.txt 2>&1; cat $OUT/cli_timing_1decoder.txt
