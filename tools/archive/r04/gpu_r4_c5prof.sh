#!/bin/bash
# round 4: BASELINE config 5 at k = 31 on the dense rows (nine-bit annots): the new test of the whole shard, rocprofv3 kernel stats of the
# bench command, separate FETCH_SIZE / WRITE_SIZE passes -> profiles/traffic.json[c5_dense]
TAG=${1:-r4c5p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_shard" 2>&1 | tail -4 | tee $OUT/pytest_c5.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c5 -- python bench.py --workload c5 --k 31 --steps 200 --warmup 20 --cpu-sample 0 --headline-only > $OUT/bench_c5_under_rocprof.json 2>> $OUT/prof.err
head -6 $OUT/prof/c5_kernel_stats.csv | cut -c1-220
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc/pmc_$c -o c5 -- python bench.py --workload c5 --k 31 --steps 3 --warmup 1 --cpu-sample 0 --headline-only > $OUT/bench_pmc.json 2>> $OUT/prof.err
done
ALG=$(python -c "import json; print(json.load(open('$OUT/bench_pmc.json'))['roofline']['algorithmic_bytes'])")
ALG_BYTES=$ALG RESULT_BYTES=2 python tools/pmc_summary.py c5_dense $OUT/pmc "sweep_conservation_halo3t_kernel" r04 > $OUT/traffic_c5.txt 2>&1; tail -25 $OUT/traffic_c5.txt
cp profiles/traffic.json $OUT/traffic.json
timeout 300 python bench.py --workload c5 --k 31 --steps 200 --warmup 20 > $OUT/bench_c5_k31.json 2>> $OUT/prof.err
python -c "
import json; j=json.load(open('$OUT/bench_c5_k31.json')); r=j['roofline']; print('c5 k=31: %.4g pos/s  %.4f ms  frac %.3f  traffic %s  parity %s' % (j['value'], j['ms_per_step'], r['frac'], r['traffic'], j['cpu_baseline']['parity_with_gpu_on_sample']))"
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*agent_info*" -delete; tail -2 $OUT/prof.err
