#!/bin/bash
TAG=${1:-r3c}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu -k "dense or sidecar or realistic or golden_one_shot or goldens_on or config3_full_size" 2>&1 | tail -8 | cut -c1-400
bash tools/gpu_r3_real.sh ${TAG}_real20 20000000 50 16
