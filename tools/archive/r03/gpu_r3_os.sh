#!/bin/bash
# round 3: the one-shot seam with the vectorised host packer -- builder parity first, then timing (SIMD on / off, thread counts)
TAG=${1:-r3os}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
nproc > $OUT/oneshot.txt; grep -m1 "model name" /proc/cpuinfo >> $OUT/oneshot.txt; grep -o -m1 "avx2\|avx512f" /proc/cpuinfo | sort -u >> $OUT/oneshot.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "builder or one_shot or region_index or host_form or sidecar or cli_bytes or integration" 2>&1 | tail -4
for simd in 1 0; do echo "MEMO_HOST_SIMD=$simd" >> $OUT/oneshot.txt
  MEMO_HOST_SIMD=$simd timeout 900 python tools/oneshot_timing.py >> $OUT/oneshot.txt 2>&1; done
for th in 16 24 48 64; do echo "MEMO_HOST_THREADS=$th" >> $OUT/oneshot.txt
  MEMO_HOST_THREADS=$th timeout 600 python tools/oneshot_timing.py --big-only >> $OUT/oneshot.txt 2>&1; done
grep -v amdgpu.ids $OUT/oneshot.txt | grep -v "^memo one-shot: 5000000\|^memo one-shot: 99999995"
