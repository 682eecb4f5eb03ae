#!/bin/bash
# the bench's own command line with the dense rows as the headline format, against the 4-byte rows, alternating
TAG=${1:-dh}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2 3; do for rows in packed dense; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --rows $rows 2>>$OUT/err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$rows', 'value %.4g ms_per_step %.4f kernel_ms %.4f median %.4f min %.4f frac %.3f ramp %s'%(j['value'], j['ms_per_step'], r['kernel_ms'], r['kernel_ms_median'], r['kernel_ms_min'], r['frac'], j['config'].get('clock_ramp')),
      '| others', [(o['rows'], round(o['kernel_ms_median'],4)) for o in j.get('other_row_formats',[])])" | tee -a $OUT/runs.txt
done; done
for rows in packed dense; do
  timeout 300 python bench.py --gpus 1 --steps 2000 --warmup 200 --cpu-sample 0 --rows $rows 2>>$OUT/err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$rows long', 'value %.4g ms_per_step %.4f kernel_ms %.4f median %.4f min %.4f frac %.3f'%(j['value'], j['ms_per_step'], r['kernel_ms'], r['kernel_ms_median'], r['kernel_ms_min'], r['frac']))" | tee -a $OUT/runs.txt
done
grep -v amdgpu.ids $OUT/err.txt | tail -3
