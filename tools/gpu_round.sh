#!/bin/bash
# full GPU test suite, bench line, short fuzz, then density / k=256 A/B of the two scatters
TAG=${1:-t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
python bench.py --steps 200 --warmup 20 > $OUT/bench.json 2>$OUT/bench.err; cat $OUT/bench.json
timeout 400 python tests/fuzz_gpu.py --seconds 240 > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt
for d in 1/100 2/100 3/100; do
  echo "== c3 k=31 density=$d packed u8" >> $OUT/ab.txt
  python tools/ab.py --workload c3 --k 31 --density $d --pack only --u8 --rounds 10 "0,0,0,0,1" "0,0,0,0,2" 2>>$OUT/err.txt >> $OUT/ab.txt
done
echo "== c3 k=256 packed u8" >> $OUT/ab.txt
python tools/ab.py --workload c3 --k 256 --pack only --u8 --rounds 10 "0,0,0,0,1" "0,0,0,0,2" "2048,4,0,0,2" "512,4,0,0,2" 2>>$OUT/err.txt >> $OUT/ab.txt
python - <<'PY' $OUT/ab.txt
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('=='): print(l.strip()); continue
    j=json.loads(l); print('  %-22s %.4f ms  (min %.4f)  frac %.3f'%(','.join(map(str,j['variant'])), j['ms_median'], j['ms_min'], j['frac_of_8TBs']))
PY
grep -v amdgpu.ids $OUT/err.txt | tail -5
