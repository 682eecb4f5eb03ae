#!/bin/bash
# GPU box: tile-width sweep + the other workloads + PMC passes.  outputs -> gpurun_out/<tag>/
TAG=${1:-sweep}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in 512 1024 2048 4096; do
  MEMO_TILE_W=$w timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>>$OUT/err.txt | sed "s/^/W=$w /" >> $OUT/c3_tile_sweep.txt
done
for wl in c2 c4 c5; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --cpu-sample 0 2>>$OUT/err.txt >> $OUT/workloads.txt
done
for k in 21 101; do
  timeout 300 python bench.py --workload c3 --k $k --steps 10 --warmup 2 --cpu-sample 0 2>>$OUT/err.txt >> $OUT/workloads.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o c3 -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/pmc_$c.json 2>> $OUT/err.txt
done
cat $OUT/c3_tile_sweep.txt $OUT/workloads.txt | cut -c1-400
ls $OUT/pmc_FETCH_SIZE; head -3 $OUT/pmc_FETCH_SIZE/*counter_collection.csv
tail -5 $OUT/err.txt
