#!/bin/bash
# round 5: the fused dense-view builder and the rule that decides when a view is built: tests, what the pass costs, its kernels
TAG=${1:-r5view}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$2" ]; then timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$2" 2>&1 | tail -15 | tee $OUT/pytest.txt | cut -c1-400; fi
for c in 0 1; do timeout 600 python tools/view_pass_timing.py --builders 0 --reps 2 --colour $c >> $OUT/view_pass.txt 2>> $OUT/view_pass.err; done; grep '"k": 31' $OUT/view_pass.txt | cut -c1-150; tail -3 $OUT/view_pass.err
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
j = json.load(open("$OUT/bench.json"))
print("value %.4g  ms/step %.4f  frac %.3f  kernel %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel"][:60]))
c = j["config"]
for kk in ("row_format_pass", "dense_format_pass", "dense_view_pass", "dense_view_place_pass"):
    print(kk, (c.get(kk) or {}).get("ms"))
print(json.dumps(j.get("resident_index_without_view"), indent=0)[:1500])
for o in j.get("other_row_formats", []):
    print("%-90s %.4f ms frac %.3f rows %d" % (o["rows"][:90], o["kernel_ms_median"], o["frac"], o["rows_read"]))
print(j["cpu_baseline"]["parity_with_gpu_on_sample"])
PY
tail -3 $OUT/bench.err
