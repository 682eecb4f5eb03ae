#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 5 --warmup 1 --cpu-sample 0 > $OUT/torchrun1.json 2> $OUT/torchrun1.err; echo "rc=$?"; cut -c1-300 $OUT/torchrun1.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 1 --steps 5 --warmup 1 --cpu-sample 0 --force-dist > $OUT/torchrun2.json 2> $OUT/torchrun2.err; echo "rc=$?"; python -c "
import json; j=json.load(open('$OUT/torchrun2.json')); print(j['value'], j['without_gather']['value'], j['gather_parity_sample'])"
grep -v "amdgpu.ids\|socket.cpp\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl" $OUT/torchrun2.err | tail -5
