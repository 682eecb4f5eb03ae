#!/bin/bash
# round 5: the order of the 4-byte rows inside a bucket under the membership planes kernels, now that a row no longer waits for the
# LDS (round 4 measured 1 % between the orders: profiles/r04_membership.txt) -- 1 start order, 3 the conservation order, 4 the membership order
TAG=${1:-r5mord}; CHUNKS=${2:-4}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
D=/tmp/real$CHUNKS
timeout 2400 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks $CHUNKS --out $D --threads 32 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"
ab() { echo "== $*" >> $OUT/ab.txt; timeout 600 python tools/ab.py "$@" >> $OUT/ab.txt 2>> $OUT/ab.err; }
for rep in 1 2; do for o in 1 3 4; do
  for k in 31 101; do ab --rows-file $D/memb.npz --membership --k $k --pack only --row-order $o --prepare --rounds 40 0,0,0; done
  for k in 31 101; do ab --workload c4 --k $k --pack only --row-order $o --prepare --rounds 40 0,0,0; done
done; done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()[:150]); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
grep -v "amdgpu.ids" $OUT/ab.err | tail -5
