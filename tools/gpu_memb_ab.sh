#!/bin/bash
# parity of the membership paths, then interleaved A/B of the membership algorithms on config 4 (packed rows)
TAG=${1:-t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed or smoke or resident or ragged or end_before or word_boundaries or many_genomes or randomized or config3 or config2 or annot" 2>&1 | tail -8 > $OUT/pytest.txt
cat $OUT/pytest.txt
for wl in "c4 31 0" "c4 21 0" "c4 31 10000000"; do read -r w k len <<< "$wl"
  echo "== $w k=$k length=$len packed" >> $OUT/ab.txt
  python tools/ab.py --workload $w --k $k --length $len --pack only --rounds 10 "0,0,3" "0,0,4" "2048,4,4" "512,4,4" "512,1,4" "256,1,4" 2>>$OUT/err.txt >> $OUT/ab.txt
done
python - <<'PY' $OUT/ab.txt
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('=='): print(l.strip()); continue
    j=json.loads(l); print('  %-22s %.4f ms  (min %.4f)  frac %.3f'%(','.join(map(str,j['variant'])), j['ms_median'], j['ms_min'], j['frac_of_8TBs']))
PY
grep -v amdgpu.ids $OUT/err.txt | tail -5
