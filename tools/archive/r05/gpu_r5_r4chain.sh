#!/bin/bash
# round 5: the radix-4 kernel's third and fourth block inside its row block (CHAIN) against the compiler's branches, same box:
# the sequence-built index at k = 101 / 128 / 200 (r4 is the launcher's choice there), parity first
TAG=${1:-r5chain}; CHUNKS=${2:-4}; LIBS=${3:-"oldcons ab"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "level_arrays or scatter or radix or packed_k or random or bucket_widths or golden or realistic or 120" 2>&1 | tail -3
timeout 100 python tests/fuzz_gpu.py --seconds 60 2>&1 | tail -1 | cut -c1-300
D=/tmp/real$CHUNKS
timeout 2400 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks $CHUNKS --out $D --threads 32 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"
for rep in 1 2; do for lib in $LIBS; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  for k in 101 128 200; do echo "== $lib cons k=$k" >> $OUT/ab.txt; MEMO_AMD_AB_LIB=$so timeout 600 python tools/ab.py --rows-file $D/cons.npz --k $k --pack only --prepare --rounds 60 0,0,0 0,0,0,0,3 >> $OUT/ab.txt 2>> $OUT/ab.err; done
done; done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()[:150]); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
grep -v "amdgpu.ids" $OUT/ab.err | tail -5
