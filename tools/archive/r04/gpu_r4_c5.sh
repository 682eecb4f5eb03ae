#!/bin/bash
# round 4: BASELINE config 5 (500 genomes, 2^25 positions per GPU) on one GPU with the gather path forced (one rank): the sweep on
# the ordered rows, the runs coding of uint16 slices (wire bytes, encode, decode) -- the inputs of tools/scaling_model.py --
# and the N > 1 control flow on the test transport (2 ranks on one GPU, gloo)
TAG=${1:-r4c5}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for k in 21 31 101; do
  timeout 600 python bench.py --workload c5 --k $k --force-dist --steps 200 --warmup 50 --cpu-sample 0 >> $OUT/c5_forced_gather.jsonl 2>> $OUT/err.txt
  timeout 600 python bench.py --workload c5 --k $k --steps 200 --warmup 50 --cpu-sample 4000000 >> $OUT/c5_single.jsonl 2>> $OUT/err.txt
done
python - <<PY
import json
for f in ("$OUT/c5_forced_gather.jsonl", "$OUT/c5_single.jsonl"):
    for ln in open(f):
        j = json.loads(ln)
        c = j["config"].get("gather_coding_choice")
        print(f.split("/")[-1], "k", j["config"]["k"], "ms/step %.4f" % j["ms_per_step"], "kernel_ms %.4f" % j["roofline"]["kernel_ms"], "frac %.3f" % j["roofline"]["frac"],
              "rows_read", j["config"]["rows_read"], j["roofline"]["kernel"][:40], (j.get("cpu_baseline") or {}).get("parity_with_gpu_on_sample"))
        if c:
            print("   sweep_ms %.4f" % c["sweep_ms"], {k: (v["wire_bytes"], round(v["decode_ms_per_slice"], 4), round(v["encode_ms"], 4)) for k, v in c["candidates"].items()})
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "two_ranks or runs16 or runs_coding" 2>&1 | tail -5
grep -v amdgpu.ids $OUT/err.txt | tail -5
