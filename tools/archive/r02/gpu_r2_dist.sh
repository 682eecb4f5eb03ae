#!/bin/bash
# bench.py's gather path on one GPU (RCCL with one rank): default plan, forced root weight, coded own slice
TAG=${1:-r2u}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for args in "--force-dist" "--force-dist --root-weight 0.5" "--force-dist --code-own-slice --nibble-gather" "--force-dist --code-own-slice --root-weight 0.25" "--force-dist --workload c4"; do
  echo "== bench.py $args" >> $OUT/dist.txt
  timeout 300 python bench.py $args --steps 20 --warmup 5 --cpu-sample 0 2>> $OUT/bench.err | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('value %.4g ms_per_step %.4f'%(j['value'], j['ms_per_step']), j.get('gather_parity_sample'), '|', j['config'].get('sharding'), '|', j['config'].get('gather_payload'))" >> $OUT/dist.txt 2>&1
done
cat $OUT/dist.txt | cut -c1-420; grep -v "amdgpu.ids\|socket.cpp\|destroy_process_group" $OUT/bench.err | tail -5
