#!/bin/bash
# round 4: the membership order of the rows (dealt over annot mod 32) against the conservation order, sustained; then the
# headline kernel with its level arrays cleared by ds_write_addtid_b32 (-DMEMO_CLEAR_ADDTID build) against ds_write_b128
TAG=${1:-r4memb}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_order or membership or memb or prepare or 120" 2>&1 | tail -8 | tee $OUT/pytest.txt
run() {  # lib workload k pack order extra...
  local lib=$1 wl=$2 k=$3 pack=$4 ord=$5; shift 5
  printf "%-10s %s k=%s %s order %s %s: " $lib $wl $k $pack $ord "$*" >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_$lib.so timeout 400 python tools/ab.py --workload $wl --k $k --pack $pack --row-order $ord --rounds 1200 "$@" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%s %.4f ms median  min %.4f  frac %.3f'%(j['variant'], j['ms_median'], j['ms_min'], j['frac_of_8TBs']), end='; ')
print()" >> $OUT/ab.txt
}
for rep in 1 2; do
for ord in 3 4 1; do
run ab c4 101 only $ord "0,0,0"
run ab c4 31 only $ord "0,0,0"
run ab c4 21 only $ord "0,0,0"
done
for lib in ab addtid_ab; do
run $lib c3 31 dense 0 --u8 "0,0,0"
run $lib c3 21 dense 0 --u8 "0,0,0"
done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
