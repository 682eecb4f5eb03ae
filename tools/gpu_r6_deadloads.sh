#!/bin/bash
# round 6: no load for a piece past the slice (the tree) against rounds 3-5's unconditional loads (deadloads = -DMEMO_LOAD_DEAD_PIECES),
# on everything else the table-driven kernel serves: all the dense rows (no view), k > 32, config 5 (nine-bit annots), five-row views
TAG=${1:-r6dl}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or config3 or config5 or six_row or golden_one_shot or randomized or resident or prepare" 2>&1 | tail -4 | tee $OUT/pytest.txt
for rep in 1 2; do for lib in ab deadloads; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  run() { echo "== $lib $*" >> $OUT/ab.txt; MEMO_AMD_AB_LIB=$so timeout 600 python tools/ab.py "$@" >> $OUT/ab.txt 2>> $OUT/ab.err; }
  run --workload c3 --k 31 --pack dense --u8 --rounds 30 "0,0,0,9"
  run --workload c3 --k 48 --pack dense --u8 --prepare --rounds 30 "0,0,0"
  run --workload c3 --k 64 --pack dense --u8 --prepare --rounds 30 "0,0,0"
  run --workload c5 --k 31 --pack dense --prepare --rounds 30 "0,0,0"
  run --workload c5 --k 21 --pack dense --prepare --rounds 30 "0,0,0"
done; done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()[:150]); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
grep -v amdgpu.ids $OUT/ab.err | tail -4
