#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, a bench line, and a rocprofv3 kernel trace.
# Usage: tools/gpu_check.sh [tag]      outputs -> gpurun_out/<tag>/
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showmeminfo vram 2>/dev/null | head -8 > $OUT/smi.txt
free -g > $OUT/host_mem.txt; nproc >> $OUT/host_mem.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $OUT/pytest_gpu.txt
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c3 -- python bench.py --steps 5 --warmup 1 --cpu-sample 0 > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof rc=$?"; cat $OUT/prof_bench.json
find $OUT/prof -name "*stats*" | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
