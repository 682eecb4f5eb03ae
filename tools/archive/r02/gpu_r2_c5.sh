#!/bin/bash
TAG=${1:-r2w}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for wl in "c5 31" "c5 101" "c5 21"; do read -r w k <<< "$wl"
  timeout 400 python bench.py --workload $w --k $k --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.jsonl
done
python - <<PY
import json
for l in open("$OUT/workloads.jsonl"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:52], "k=%d"%j["config"]["k"], "| %s B rows: %.3f ms (median %.3f) frac %.3f val %.3g"%(j["config"]["row_bytes"], r["kernel_ms"], r["kernel_ms_median"], r["frac"], j["value"]), r["kernel"][:60],
          "| others:", ["%s %.3f ms frac %.3f"%(o["rows"], o["kernel_ms_median"], o["frac"]) for o in j.get("other_row_formats", [])])
PY
grep -v "amdgpu.ids" $OUT/bench.err | tail -3
