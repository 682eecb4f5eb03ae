#!/bin/bash
# sidecar fast path: tests, CLI timing, then a fuzz leg
TAG=${1:-fp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "sidecar or cli or builder or import" 2>&1 | tail -6 | tee $OUT/pytest_fast.txt | cut -c1-400
timeout 600 python tools/cli_timing.py --out /tmp/cli > $OUT/cli_timing.txt 2>&1; tail -25 $OUT/cli_timing.txt | cut -c1-300
timeout 300 python tools/oneshot_timing.py > $OUT/oneshot.txt 2>&1; tail -12 $OUT/oneshot.txt | cut -c1-300
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt | cut -c1-300
timeout 420 python tests/fuzz_gpu.py --seconds 300 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
