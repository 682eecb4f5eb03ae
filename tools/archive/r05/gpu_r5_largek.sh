#!/bin/bash
# round 5, VERDICT r04 item 2: what the large-k conservation sweep and the membership sweep on the sequence-built index are short of --
# A/B in one process per library (tools/ab.py: interleaved, results checked), the diagnostic builds of tools/build_variant.sh beside it
#   gpu_r5_largek.sh TAG CHUNKS "lib1 lib2 ..."      (libraries: memo_amd/libmemo_amd_<name>_ab.so; "ab" = the product's A/B build)
TAG=${1:-r5lk}; CHUNKS=${2:-4}; LIBS=${3:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
D=/tmp/real$CHUNKS
timeout 2400 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks $CHUNKS --out $D --threads 32 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"
for lib in $LIBS; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  for k in 101 64; do
    echo "== $lib cons k=$k" | tee -a $OUT/ab.txt
    MEMO_AMD_AB_LIB=$so timeout 600 python tools/ab.py --rows-file $D/cons.npz --k $k --pack only --rounds 40 ${CONS_VARIANTS:-"0,0,0"} >> $OUT/ab.txt 2>> $OUT/ab.err
  done
  if [ -n "$MEMB_VARIANTS" ]; then
    for k in 31 101; do
      echo "== $lib memb k=$k" | tee -a $OUT/ab.txt
      MEMO_AMD_AB_LIB=$so timeout 600 python tools/ab.py --rows-file $D/memb.npz --membership --k $k --pack only --rounds 40 $MEMB_VARIANTS >> $OUT/ab.txt 2>> $OUT/ab.err
    done
  fi
done
if [ -n "$C4_VARIANTS" ]; then
  for k in 101 31; do
    echo "== ab c4 k=$k" | tee -a $OUT/ab.txt
    timeout 600 python tools/ab.py --workload c4 --k $k --pack only --rounds 40 $C4_VARIANTS >> $OUT/ab.txt 2>> $OUT/ab.err
  done
fi
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
grep -v "amdgpu.ids" $OUT/ab.err | tail -5
