#!/bin/bash
# end-to-end `memo query` phases on a synthetic Parquet index, for several row-group decoder counts
TAG=${1:-cli}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/nproc.txt
for th in 8 16 32 1; do
  MEMO_DECODE_THREADS=$th timeout 900 python tools/cli_timing.py --num-docs 100 --pivot 20000000 --out /tmp/cli_t > $OUT/cli_timing_${th}decoders.txt 2>&1
  echo "== $th decoders"; grep -v "^wrote" $OUT/cli_timing_${th}decoders.txt | head -12
done
