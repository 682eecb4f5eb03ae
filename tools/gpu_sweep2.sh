#!/bin/bash
# GPU box: parity tests, then variant sweeps (waves x tile width, membership algorithms), force-dist check.
TAG=${1:-sweep2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
run() { # label, env..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 "$@" 2>>$OUT/err.txt | sed "s/^/$label /" >> $OUT/results.txt
}
for w in 1024 2048 4096; do run "cons_w4_W$w" MEMO_WAVES=4 MEMO_TILE_W=$w -- ; done
for w in 512 1024; do run "cons_w1_W$w" MEMO_WAVES=1 MEMO_TILE_W=$w -- ; done
for w in 256 512 1024; do run "memb_dbl_w4_W$w" MEMO_WAVES=4 MEMO_MEMB_ALGO=2 MEMO_TILE_W=$w -- --workload c4; done
for w in 256 512; do run "memb_dbl_w1_W$w" MEMO_WAVES=1 MEMO_MEMB_ALGO=2 MEMO_TILE_W=$w -- --workload c4; done
for w in 512 1024 2048; do run "memb_dir_w1_W$w" MEMO_WAVES=1 MEMO_MEMB_ALGO=1 MEMO_TILE_W=$w -- --workload c4; done
for w in 2048 4096; do run "memb_dir_w4_W$w" MEMO_WAVES=4 MEMO_MEMB_ALGO=1 MEMO_TILE_W=$w -- --workload c4; done
run "k101_w1_W512" MEMO_WAVES=1 MEMO_TILE_W=512 -- --k 101
run "k101_w4_W2048" MEMO_WAVES=4 MEMO_TILE_W=2048 -- --k 101
run "k101_w4_W4096" MEMO_WAVES=4 MEMO_TILE_W=4096 -- --k 101
run "default_c4" -- --workload c4
run "forcedist_c3" -- --force-dist
run "forcedist_c3_wide" -- --force-dist --wide
python - <<PY
import json
for line in open("$OUT/results.txt"):
    lab, js = line.split(" ", 1)
    j = json.loads(js); r = j["roofline"]
    print("%-18s kern=%.3fms step=%.3fms frac=%.3f val=%.3g %s" % (lab, r["kernel_ms"], j["ms_per_step"], r["frac"], j["value"], j.get("gather_parity_sample", "")))
PY
tail -5 $OUT/err.txt
