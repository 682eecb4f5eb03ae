#!/bin/bash
# why is the seam slower inside bench.py than in tools/oneshot_sweep.py?  the library's phase lines of both
TAG=${1:-r6seam}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
MEMO_TIMING=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
grep "memo one-shot" $OUT/bench.err
python - <<PY
import json
j = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(json.dumps(j["one_shot_seam"])[:600])
PY
echo "== standalone"
timeout 600 python tools/oneshot_sweep.py 3 2>&1 | grep -v amdgpu.ids
