#!/bin/bash
# kernel durations of the transport codings inside the bench's gather path (one rank over RCCL)
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "transport" 2>&1 | tail -25 > $OUT/pytest.txt; tail -25 $OUT/pytest.txt
for mode in "" "--nibble-gather" "--coding=runs"; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$mode -o t -- python bench.py --force-dist --code-own-slice --steps 10 --warmup 2 --cpu-sample 0 $mode > $OUT/t$mode.json 2>> $OUT/prof.err
  echo "== $mode"; cut -d, -f1-4 $OUT/prof$mode/t_kernel_stats.csv | head -12
done
