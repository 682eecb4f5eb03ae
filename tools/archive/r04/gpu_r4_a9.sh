#!/bin/bash
# round 4: BASELINE config 5 (500 genomes) on the dense rows (nine-bit annots) against its 4-byte rows; sustained, one format per process
TAG=${1:-r4a9}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  for k in 31 21; do
    for pk in only dense; do
      echo -n "c5 k=$k $pk: " >> $OUT/ab.txt
      python tools/ab.py --workload c5 --k $k --pack $pk --rounds 1500 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f  rows_read %d  sweep %d' % (j['ms_median'], j['ms_min'], j['last_rows_read'], j['last_sweep']))" >> $OUT/ab.txt
    done
  done
done
sort $OUT/ab.txt
for k in 31 21; do
  python bench.py --workload c5 --k $k --steps 300 --warmup 50 > $OUT/bench_c5_k$k.json 2>$OUT/bench_c5_k$k.err
  python -c "
import json; j=json.loads(open('$OUT/bench_c5_k$k.json').read().strip().splitlines()[-1])
print('c5 k=$k', j['config']['row_format'][:40], '%.4g pos/s  %.4f ms  frac %.3f  parity %s' % (j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline'].get('parity_with_gpu_on_sample')))
for o in j.get('other_row_formats', []): print('   other:', o['rows'], '%.4f ms' % o['kernel_ms_median'])"
done
tail -2 $OUT/err.txt
