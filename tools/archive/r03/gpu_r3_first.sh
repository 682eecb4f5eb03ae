#!/bin/bash
# round 3, first GPU pass: the new multi-device test, bench.py plain / self-launched / forced RCCL path at N = 1
TAG=${1:-r3a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "multi_device" 2>&1 | tail -4 | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.json 2> $OUT/bench_plain.err; echo "plain rc=$?"; cut -c1-300 $OUT/bench_plain.json
timeout 600 python bench.py --gpus 1 --launch --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_launch.json 2> $OUT/bench_launch.err; echo "launch rc=$?"; cut -c1-300 $OUT/bench_launch.json
timeout 600 python bench.py --gpus 1 --launch --force-dist --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_dist.json 2> $OUT/bench_dist.err; echo "dist rc=$?"; cut -c1-300 $OUT/bench_dist.json
timeout 600 python bench.py --gpus 1 --launch --force-dist --plain-gather --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_dist_plain.json 2> $OUT/bench_dist_plain.err; echo "dist plain rc=$?"; cut -c1-300 $OUT/bench_dist_plain.json
timeout 60 python bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_two.json 2> $OUT/bench_two.err; echo "two rc=$? (expected 2)"; tail -1 $OUT/bench_two.err
python - <<PY
import json
a=json.load(open("$OUT/bench_plain.json")); 
for f in ("bench_launch","bench_dist","bench_dist_plain"):
    try:
        b=json.load(open("$OUT/%s.json"%f)); print(f, "value %.4g vs plain %.4g: %+.2f %%"%(b["value"], a["value"], 100*(b["value"]/a["value"]-1)), b.get("link_probe"), (b.get("ranks_seen") or {}).get("distinct_devices"))
    except Exception as e: print(f, "failed", e)
PY
grep -v "amdgpu.ids\|socket.cpp\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl" $OUT/bench_dist.err | tail -5
