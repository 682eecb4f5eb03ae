#!/bin/bash
# parity of the packed-row tests, then interleaved A/B of the clipped (1) and unclipped (2) scatter
TAG=${1:-t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packed or smoke or resident or ragged or end_before" 2>&1 | tail -8 > $OUT/pytest.txt
cat $OUT/pytest.txt
for wl in "c3 31 0" "c3 21 0" "c3 101 0" "c3 31 1000000" "c3 31 10000000" "c3 31 30000000" "c5 31 0" "c5 101 0"; do read -r w k len <<< "$wl"
  echo "== $w k=$k length=$len packed u8" >> $OUT/ab.txt
  U8="--u8"; [ $w = c5 ] && U8=""
  python tools/ab.py --workload $w --k $k --length $len --pack only $U8 --rounds 12 "0,0,0,0,1" "0,0,0,0,2" "1024,4,0,0,2" "512,1,0,0,2" "512,4,0,0,2" "2048,4,0,0,2" 2>>$OUT/err.txt >> $OUT/ab.txt
done
for lib in w8u8 w8u5; do
  echo "== lib $lib c3 k=31 packed u8" >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_$lib.so python tools/ab.py --workload c3 --k 31 --pack only --u8 --rounds 12 "0,0,0,0,1" "0,0,0,0,2" 2>>$OUT/err.txt >> $OUT/ab.txt
done
python - <<'PY' $OUT/ab.txt
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('=='): print(l.strip()); continue
    j=json.loads(l); print('  %-22s %.4f ms  (min %.4f)  frac %.3f'%(','.join(map(str,j['variant'])), j['ms_median'], j['ms_min'], j['frac_of_8TBs']))
PY
grep -v amdgpu.ids $OUT/err.txt | tail -5
