#!/bin/bash
# the packed, pinned way in: its tests, then the one-shot timing with phases
TAG=${1:-r2b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "builder or golden or cli or region_index or end_to_end or sharded or dap_to_parquet or randomized" 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt
nproc > $OUT/oneshot.txt; free -g | head -2 >> $OUT/oneshot.txt
timeout 900 python tools/oneshot_timing.py >> $OUT/oneshot.txt 2>&1; grep -v amdgpu.ids $OUT/oneshot.txt
for t in 8 16 32 64 96; do echo "MEMO_HOST_THREADS=$t" >> $OUT/threads.txt; MEMO_HOST_THREADS=$t timeout 600 python tools/oneshot_timing.py 2>&1 | grep "N=100 L=100000000\|499999995 rows packed" | tail -3 >> $OUT/threads.txt; done; cat $OUT/threads.txt
