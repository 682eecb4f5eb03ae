#!/bin/bash
# round 6: no load for a piece past the slice on sparse tiles (the tree: the launcher picks the SP kernels) against rounds 3-5's loads
# everywhere (spnever = -DMEMO_SPARSE_NEVER): parity, the view sweeps by k, bench legs incl. config 5
TAG=${1:-r6sp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense or config3 or config5 or six_row or golden_one_shot or randomized or resident or prepare or views or level or multi_device" 2>&1 | tail -4 | tee $OUT/pytest.txt
bash tools/gpu_r6_prologue.sh $TAG "ab spnever"
LIBS="ab spnever" bash tools/gpu_r6_deadloads2.sh $TAG
