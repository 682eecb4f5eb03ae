#!/bin/bash
# round 6, VERDICT r05 item 3: what each phase of the headline kernel is worth TODAY (six-row placed views), by ablation builds
# (tools/build_variant.sh ablN -DMEMO_T_ABLATE=N: 1 rows loaded and dropped, 2 no clear, 4 no fold, 6 neither; results are wrong,
# they size a phase), same box, one variant per process, sustained
TAG=${1:-r6head}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ab abl2 abl4 abl6 abl1; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  [ -f $so ] || continue
  echo "== $lib" >> $OUT/head.txt
  MEMO_AMD_AB_LIB=$so timeout 600 python tools/view_sweep_ab.py --ks 31,21,17,9 --variants 0:6:1 --reps 1 --launches 600 >> $OUT/head.txt 2>> $OUT/head.err
done; done
python3 - <<PY
import json
cur=None
for l in open("$OUT/head.txt"):
    if l.startswith("=="): cur=l.strip(); print(cur); continue
    j=json.loads(l); print("   k=%-3d %.4f ms (min %.4f) rows %d variant %d" % (j["k"], j["ms_median"], j["ms_min"], j["rows_read"], j["variant"]))
PY
grep -v amdgpu.ids $OUT/head.err | tail -3
