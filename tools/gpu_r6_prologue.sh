#!/bin/bash
# round 6: the headline kernel's prologue -- kernel arguments in one scalar round trip (early), the clear under the descriptor's
# latency (clearfirst), both (earlycf) -- against the tree (ab); same box, one variant per process, sustained, two rounds
TAG=${1:-r6pro}; LIBS=${2:-"ab clearfirst early earlycf"}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do for lib in $LIBS; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  [ -f $so ] || continue
  echo "== $lib" >> $OUT/head.txt
  MEMO_AMD_AB_LIB=$so timeout 600 python tools/view_sweep_ab.py --ks 31,21,17,9 --variants 0:6:1 --reps 1 --launches 600 >> $OUT/head.txt 2>> $OUT/head.err
done; done
python3 - <<PY
import json, collections
cur=None; acc=collections.defaultdict(list)
for l in open("$OUT/head.txt"):
    if l.startswith("=="): cur=l.strip()[3:]; continue
    j=json.loads(l); acc[(cur, j["k"])].append(j["ms_median"])
libs=[]
for (lib,k) in acc:
    if lib not in libs: libs.append(lib)
for lib in libs:
    print("%-12s" % lib, "  ".join("k=%d %s" % (k, " ".join("%.4f" % x for x in acc[(lib,k)])) for k in (31,21,17,9)))
PY
grep -v amdgpu.ids $OUT/head.err | tail -3
