#!/bin/bash
# round 5, item 2, second run: membership planes with the hand-written row block / byte-wise transposes / tile width by result words
# (parity first), then A/B on the sequence-built index and on config 4; the r4 kernel's second-block ablations at k = 101
TAG=${1:-r5lk2}; CHUNKS=${2:-4}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "memb or plane or golden or cli" 2>&1 | tail -5 | tee $OUT/pytest.txt | cut -c1-300
timeout 100 python tests/fuzz_gpu.py --seconds 45 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
D=/tmp/real$CHUNKS
timeout 2400 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks $CHUNKS --out $D --threads 32 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"
ab() { echo "== $*" >> $OUT/ab.txt; timeout 600 python tools/ab.py "$@" >> $OUT/ab.txt 2>> $OUT/ab.err; }
for k in 31 101; do ab --rows-file $D/memb.npz --membership --k $k --pack only --rounds 40 0,0,0 1024,4,0 2048,4,0 4096,4,0 1024,1,0; done
for k in 31 101; do ab --workload c4 --k $k --pack only --rounds 40 0,0,0 2048,4,0 512,4,0; done
ab --workload c4 --num-docs 30 --k 31 --pack only --rounds 40 0,0,0 1024,4,0 2048,4,0
ab --workload c4 --num-docs 250 --k 31 --pack only --rounds 40 0,0,0 512,4,0
for lib in ab abl32 abl64; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  for rep in 1 2; do echo "== $lib cons k=101" >> $OUT/ab.txt; MEMO_AMD_AB_LIB=$so timeout 600 python tools/ab.py --rows-file $D/cons.npz --k 101 --pack only --rounds 60 0,0,0 >> $OUT/ab.txt 2>> $OUT/ab.err; done
done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()[:150]); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
grep -v "amdgpu.ids" $OUT/ab.err | tail -5
