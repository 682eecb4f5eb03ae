#!/bin/bash
# round 4: the weakest line of the large index from sequences, k = 101 (0.385 of 8 TB/s): level arrays and array sizes on rows
# built from sequences (2 x 20 Mbp x 50 genomes: 1.9e8 rows, 0.76 GB of 4-byte rows -- past the Infinity Cache), sustained A/B in
# one process per variant: tile_w,waves,memb,row_source,scatter  (scatter 2 doubling, 3 radix-4, 4 mixed all arrays, 5 mixed planned)
TAG=${1:-r4real2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks 2 --out /tmp/real2 --threads 32 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"
run() {  # k pack extra...
  local k=$1 pack=$2; shift 2
  printf "k=%s %s %s: " $k $pack "$*" >> $OUT/ab.txt
  timeout 400 python tools/ab.py --rows-file /tmp/real2/cons.npz --k $k --pack $pack --rounds 800 "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%s %.4f ms median  min %.4f  sweep %d arrays %d rows_read %d'%(j['variant'], j['ms_median'], j['ms_min'], j['last_sweep'], j['level_arrays'], j['last_rows_read']), end='; ')
print()" >> $OUT/ab.txt
}
for rep in 1 2; do
for k in 65 72 80 101 128 160 200 256; do run $k only --u8 "0,0,0"; done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_realistic_index.py -x -q -m gpu -k "level_arrays or full_size or realistic or config5" 2>&1 | tail -5 | tee $OUT/pytest.txt
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
