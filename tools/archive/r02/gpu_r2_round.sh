#!/bin/bash
# full tier + fuzz + the round's evidence run (tools/gpu_r2_profile.sh) on one box
TAG=${1:-rr}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python -c "
from memo_amd import _lib
print('library first: devices', _lib.lib().memo_device_count())
import torch
print('then torch: cuda available', torch.cuda.is_available())" > $OUT/load_order.txt 2>&1
python -c "
import torch
print('torch first: cuda available', torch.cuda.is_available())
from memo_amd import _lib
print('then library: devices', _lib.lib().memo_device_count())" >> $OUT/load_order.txt 2>&1
grep -v amdgpu.ids $OUT/load_order.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "transport or multi or cli" 2>&1 | tail -3 | tee $OUT/pytest_alone.txt | cut -c1-300
timeout 300 python tests/fuzz_gpu.py --seconds 180 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-300
bash tools/gpu_r2_profile.sh $TAG/prof
