#!/bin/bash
# round 4: the GPU tier, the smoke, bench.py on the driver's command line
TAG=${1:-r4tier}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; tail -3 $OUT/bench_driver.err
python -c "
import json; j=json.load(open('$OUT/bench_driver.json'))
print('value %.4g  ms/step %.4f  frac %.3f  kernel %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel'][:60]))
print('cpu parity', j['cpu_baseline']['parity_with_gpu_on_sample'], ' view_pass', j['config']['dense_view_pass'])
for o in j.get('other_row_formats', []): print('  other:', o['rows'], '%.4f ms frac %.3f' % (o['kernel_ms_median'], o['frac']))
print('pack pass', j['config']['row_format_pass']['ms'] if j['config']['row_format_pass'] else None)
"
