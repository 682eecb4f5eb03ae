#!/bin/bash
# round 6: `memo query` on BASELINE config 3 itself (5e8-row Parquet, chr1:0-100000000 conservation, 1e7-position membership)
TAG=${1:-r6cli}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
{ nproc; cat /sys/fs/cgroup/cpu.max; df -h /tmp | tail -1; } > $OUT/cli_timing_c3.txt
timeout 2400 python tools/cli_timing.py --num-docs 100 --pivot 100000000 --memb-window 10000000 --out /tmp/cli_c3 >> $OUT/cli_timing_c3.txt 2>&1
cat $OUT/cli_timing_c3.txt
