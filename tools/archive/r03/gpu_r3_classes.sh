#!/bin/bash
# round 3: k-class views in classes of four -- tests, then the workloads at k = 21 (the k that gained: cap 20 instead of 32)
TAG=${1:-r3cl}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "k_class or no_room or dense_row_sweep or config5 or planes or memb" 2>&1 | tail -4
: > $OUT/w.jsonl
timeout 400 python bench.py --workload c3 --k 21 --steps 50 --warmup 10 --cpu-sample 0 --headline-only >> $OUT/w.jsonl 2>> $OUT/err.txt
timeout 400 python bench.py --workload c5 --k 21 --steps 50 --warmup 10 --cpu-sample 0 --headline-only >> $OUT/w.jsonl 2>> $OUT/err.txt
timeout 400 python bench.py --workload c4 --k 21 --steps 50 --warmup 10 --cpu-sample 0 --headline-only >> $OUT/w.jsonl 2>> $OUT/err.txt
timeout 400 python bench.py --workload c3 --k 31 --steps 50 --warmup 10 --cpu-sample 0 --headline-only >> $OUT/w.jsonl 2>> $OUT/err.txt
python - <<PY
import json
for l in open("$OUT/w.jsonl"):
    j = json.loads(l); r = j["roofline"]; c = j["config"]
    print(c["workload"][:18], "k", c["k"], c["query"], "rows_read", c["rows_read"], "%.4f ms" % r["kernel_ms"], "%.3g pos/s" % j["value"], "frac %.3f" % r["frac"])
PY
grep -v amdgpu.ids $OUT/err.txt | tail -3
