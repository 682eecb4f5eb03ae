#!/bin/bash
# round 6: the same A/B through bench.py's own sustained legs (settle + >= 100 launches each): headline, all dense rows, five-row view,
# uint16 results; config 5 at k = 31 / 21; config 3 at k = 64 (no view: all rows).  MEMO_AMD_LIB selects the library.
TAG=${1:-r6dl2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do for lib in ${LIBS:-ab deadloads}; do
  so=$PWD/memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=$PWD/memo_amd/libmemo_amd_ab.so
  MEMO_AMD_LIB=$so timeout 600 python bench.py --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/err.txt | sed "s/^/$lib c3k31 /" >> $OUT/lines.txt
  for wl in "c5 31" "c5 21" "c3 64" "c3 48"; do read -r w k <<< "$wl"
    MEMO_AMD_LIB=$so timeout 600 python bench.py --workload $w --k $k --steps 100 --warmup 20 --cpu-sample 0 --headline-only 2>>$OUT/err.txt | sed "s/^/$lib ${w}k$k /" >> $OUT/lines.txt
  done
done; done
python3 - <<PY
import json
for l in open("$OUT/lines.txt"):
    lib, wl, js = l.split(" ", 2)
    j = json.loads(js); r = j["roofline"]
    s = "%-10s %-7s headline %.4f (median %.4f)" % (lib, wl, r["kernel_ms"], r["kernel_ms_median"])
    for o in j.get("other_row_formats", []):
        if "all rows" in o["rows"] or "five rows" in o["rows"] or "uint16" in o["rows"]:
            s += " | %s %.4f" % (o["rows"][:28], o["kernel_ms_median"])
    print(s)
PY
grep -v amdgpu.ids $OUT/err.txt | tail -3
