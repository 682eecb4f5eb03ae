#!/bin/bash
# round 6: the one-shot seam by host thread count (the box's cgroup grants 16 CPUs of 256), both first-touch patterns
TAG=${1:-r6os}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
{
cat /sys/fs/cgroup/cpu.max
timeout 900 python -m pytest tests -m gpu -x -q -k "one_shot or builder or golden_one_shot or import or export or unpackable" 2>&1 | tail -5
for t in "" 8 12 16 24 32 64 128; do
  echo "== MEMO_HOST_THREADS=${t:-default}"
  if [ -z "$t" ]; then timeout 600 python tools/oneshot_sweep.py 4; else MEMO_HOST_THREADS=$t timeout 600 python tools/oneshot_sweep.py 3; fi
done
} > $OUT/oneshot.txt 2>&1
grep -v amdgpu.ids $OUT/oneshot.txt | tail -150
