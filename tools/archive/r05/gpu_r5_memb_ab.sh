#!/bin/bash
# round 5, item 2(b): the membership planes kernels before and after (libraries named on the command line; "ab" = the tree's), same box:
# the sequence-built index (two result words) and config 4 (four) with its k-class view prepared
TAG=${1:-r5mab}; CHUNKS=${2:-4}; LIBS=${3:-"oldmemb ab"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
D=/tmp/real$CHUNKS
timeout 2400 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks $CHUNKS --out $D --threads 32 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"
for rep in 1 2; do for lib in $LIBS; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  ab() { echo "== $lib $*" >> $OUT/ab.txt; MEMO_AMD_AB_LIB=$so timeout 600 python tools/ab.py "$@" >> $OUT/ab.txt 2>> $OUT/ab.err; }
  for k in ${REAL_KS:-31 101}; do ab --rows-file $D/memb.npz --membership --k $k --pack only --prepare --rounds 40 1024,4,0 2048,4,0; done
  for k in ${C4_KS:-31 21 101}; do ab --workload c4 --k $k --pack only --prepare --rounds 40 1024,4,0; done
done; done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()[:150]); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
grep -v "amdgpu.ids" $OUT/ab.err | tail -5
