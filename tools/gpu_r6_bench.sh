#!/bin/bash
# round 6: the split bench.py -- the N = 1 line as the driver runs it, its keys against round 5's, the N > 1 control flow on the test transport
TAG=${1:-r6b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err
python - <<PY
import json
new = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
old = json.load(open("BENCH_r05.json"))["parsed"]
def keys(d, p=""):
    out = set()
    for k, v in d.items():
        out.add(p + k)
        if isinstance(v, dict): out |= keys(v, p + k + ".")
    return out
print("keys gone:", sorted(keys(old) - keys(new)))
print("keys new :", sorted(keys(new) - keys(old)))
print("value %.4g  ms_per_step %.5f  frac %.3f  kernel_ms %.5f" % (new["value"], new["ms_per_step"], new["roofline"]["frac"], new["roofline"]["kernel_ms"]))
print("seam", json.dumps(new.get("one_shot_seam"), indent=None)[:1500])
print("cpu_baseline", json.dumps(new.get("cpu_baseline"))[:900])
PY
timeout 2400 python -m pytest tests -x -q -m gpu -k "bench or cli or front_end or cache or golden_one_shot or integration_stub" 2>&1 | tail -8
