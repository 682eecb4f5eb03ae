#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "transport" 2>&1 | tail -2
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 5 --warmup 1 --cpu-sample 0 > $OUT/torchrun1.json 2> $OUT/torchrun1.err; echo "rc=$?"; cut -c1-200 $OUT/torchrun1.json
for extra in "" "--coding runs" "--coding runs --k 101" "--nibble-gather" "--plain-gather" "--workload c5" "--workload c4" "--wide" "--k 101" "--workload c2"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 --force-dist --code-own-slice $extra > $OUT/t.json 2> $OUT/t.err; echo "rc=$? [$extra]"; python -c "
import json; j=json.load(open('$OUT/t.json')); print('%.3g with gather, %.3g sweeps only'%(j['value'], j['without_gather']['value']), j['gather_parity_sample'], j['config'].get('gather_payload'))"
  grep -v "amdgpu.ids\|socket.cpp\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl" $OUT/t.err | tail -3
done
