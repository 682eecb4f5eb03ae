#!/bin/bash
# round 6: what the driver runs at round end -- smoke, the whole GPU tier, the bench line
TAG=${1:-r6final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python3 - <<PY
import json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
s = j["one_shot_seam"]
print("value %.4g ms_per_step %.5f frac %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["frac"]))
print("seam", s.get("ms_calls"), "first", s.get("first_call_ms"), "pcie", s.get("pcie"), s.get("error"))
PY
