#!/bin/bash
TAG=${1:-r3pv}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu -k "views or packed or config5 or config3_full or level_arrays or resident_index or golden_one_shot or sidecar" 2>&1 | tail -6 | cut -c1-400
timeout 200 python tests/fuzz_gpu.py --seconds 100 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-500
for wl in "c4 31" "c5 31" "c5 21" "c3 31 --rows packed" "c4 21"; do read -r w k extra <<< "$wl"
  timeout 400 python bench.py --workload $w --k $k --steps 200 --warmup 20 --cpu-sample 0 $extra 2>>$OUT/bench.err >> $OUT/workloads.jsonl
done
python - <<PY
import json
for l in open("$OUT/workloads.jsonl"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:52], "k=%d"%j["config"]["k"], "| %s B rows, read %s: %.3f ms (median %.3f) frac %.3f val %.3g"%(j["config"]["row_bytes"], j["config"].get("rows_read"), r["kernel_ms"], r["kernel_ms_median"], r["frac"], j["value"]),
          "| others:", ["%s %.3f ms frac %.3f"%(o["rows"][:12], o["kernel_ms_median"], o["frac"]) for o in j.get("other_row_formats", [])])
PY
grep -v "amdgpu.ids" $OUT/bench.err | tail -3
