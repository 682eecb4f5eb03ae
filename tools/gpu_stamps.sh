#!/bin/bash
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_stamps.so
for args in "--workload c3 --pack only" "--workload c3 --pack only --k 101" "--workload c3" "--workload c5 --pack only" "--workload c3 --pack only --tuning 256,1,0"; do
  python tools/stamps.py $args >> $OUT/stamps.txt 2>>$OUT/err.txt
done
cat $OUT/stamps.txt; grep -v amdgpu.ids $OUT/err.txt | tail -3
