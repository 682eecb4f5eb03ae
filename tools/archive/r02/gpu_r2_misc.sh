#!/bin/bash
# new tests of the last edits + transport kernel timing + single-rank RCCL gather path of bench.py
TAG=${1:-r2m}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "transport or builder or integration_stub or end_before or sidecar" 2>&1 | tail -6 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_force_dist.json 2> $OUT/bench.err; echo "rc=$?"
python - <<PY
import json
j=json.load(open("$OUT/bench_force_dist.json"))
print("force-dist: value %.3g ms_per_step %.3f"%(j["value"], j["ms_per_step"]), j.get("gather_parity_sample"), j["config"].get("gather_payload"))
c=j["config"].get("gather_coding_choice")
if c:
    for k,v in c["candidates"].items(): print("  ", k, {a: round(b,4) if isinstance(b,float) else b for a,b in v.items()})
PY
timeout 300 python bench.py --force-dist --code-own-slice --nibble-gather --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_force_dist_nibble.json 2>> $OUT/bench.err
python - <<PY
import json
j=json.load(open("$OUT/bench_force_dist_nibble.json"))
print("force-dist nibble own slice: value %.3g ms_per_step %.3f"%(j["value"], j["ms_per_step"]), j.get("gather_parity_sample"), j["config"].get("gather_payload"))
PY
grep -v "amdgpu.ids\|socket.cpp" $OUT/bench.err | tail -5
