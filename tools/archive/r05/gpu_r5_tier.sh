#!/bin/bash
# round 5: smoke, the whole GPU tier, a minute of fuzz, the bench line on the driver's command line
TAG=${1:-r5tier}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt | cut -c1-400
timeout 120 python tests/fuzz_gpu.py --seconds ${2:-60} > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
j = json.load(open("$OUT/bench_driver.json"))
print("value %.4g  ms/step %.4f  frac %.3f  kernel %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel"][:60]))
c = j["config"]
for kk in ("row_format_pass", "dense_format_pass", "dense_view_pass", "dense_view_place_pass"):
    print(kk, (c.get(kk) or {}).get("ms"))
r = j.get("resident_index_without_view") or {}
print({k: r.get(k) for k in ("kernel_ms_median", "view_build_ms", "view_amortised_after_queries", "places_build_ms", "places_amortised_after_queries")})
print("one_shot_seam", j.get("one_shot_seam"))
for o in j.get("other_row_formats", []):
    print("%-90s %.4f ms frac %.3f rows %d" % (o["rows"][:90], o["kernel_ms_median"], o["frac"], o["rows_read"]))
print("cpu parity", j["cpu_baseline"]["parity_with_gpu_on_sample"], "cpu 1 core %.3g all %.3g" % (j["cpu_baseline"]["value"], j["cpu_baseline"]["all_cores"]["value"]))
PY
grep -v amdgpu.ids $OUT/bench.err | tail -3
