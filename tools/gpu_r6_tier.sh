#!/bin/bash
# round 6: the whole GPU tier, then the membership row blocks' EXEC ending A/B (oldexec = `s_mov_b64 exec, -1`, ab = save / restore), same box
TAG=${1:-r6tier}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
if [ -f memo_amd/libmemo_amd_oldexec_ab.so ]; then
for rep in 1 2; do for lib in oldexec ab; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  for k in 31 21 101; do echo "== $lib c4 k=$k" >> $OUT/ab.txt; MEMO_AMD_AB_LIB=$so timeout 600 python tools/ab.py --workload c4 --k $k --pack only --prepare --rounds 40 1024,4,0 >> $OUT/ab.txt 2>> $OUT/ab.err; done
done; done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()[:150]); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python3 - <<PY
import json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
s = j["one_shot_seam"]
print("value %.4g ms_per_step %.5f frac %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"]))
print("seam", s.get("ms_calls"), "first", s.get("first_call_ms"), "one-thread", s.get("columns_first_touched_by_one_thread", {}).get("ms_calls"), "pcie", s.get("pcie"), "floor", s.get("floor_ms_from_pcie"), s.get("error"))
PY
