#!/bin/bash
TAG=${1:-pack2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
V='256,1,0 512,1,0 1024,1,0 256,4,0 512,4,0 1024,4,0 2048,4,0'
python tools/ab.py --workload c3 --k 31 --pack only $V > $OUT/ab_c3_packed_u4.txt 2>$OUT/err.txt
MEMO_AMD_LIB=$PWD/memo_amd/libmemo_amd_u2.so python tools/ab.py --workload c3 --k 31 --pack only $V > $OUT/ab_c3_packed_u2.txt 2>>$OUT/err.txt
MEMO_AMD_LIB=$PWD/memo_amd/libmemo_amd_u8.so python tools/ab.py --workload c3 --k 31 --pack only $V > $OUT/ab_c3_packed_u8.txt 2>>$OUT/err.txt
MEMO_AMD_LIB=$PWD/memo_amd/libmemo_amd_u8.so python tools/ab.py --workload c3 --k 31 "1024,1,0" "4096,4,0" "2048,4,0" > $OUT/ab_c3_wide_u8.txt 2>>$OUT/err.txt
MEMO_AMD_LIB=$PWD/memo_amd/libmemo_amd_u2.so python tools/ab.py --workload c3 --k 31 "1024,1,0" "4096,4,0" "2048,4,0" > $OUT/ab_c3_wide_u2.txt 2>>$OUT/err.txt
python tools/ab.py --workload c4 --k 31 --pack only "256,4,2" "256,1,2" "512,4,2" > $OUT/ab_c4_packed_u4.txt 2>>$OUT/err.txt
MEMO_AMD_LIB=$PWD/memo_amd/libmemo_amd_u2.so python tools/ab.py --workload c4 --k 31 --pack only "256,4,2" "256,1,2" "512,4,2" > $OUT/ab_c4_packed_u2.txt 2>>$OUT/err.txt
for f in $OUT/ab_*.txt; do echo "== $f"; python - "$f" <<PY
import json,sys
for l in open(sys.argv[1]):
    j=json.loads(l); print(j["variant"], "%.3f ms  frac %.3f"%(j["ms_median"], j["frac_of_8TBs"]))
PY
done; grep -v "amdgpu.ids" $OUT/err.txt | tail -5
