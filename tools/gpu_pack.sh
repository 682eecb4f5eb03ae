#!/bin/bash
TAG=${1:-pack}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
python tools/ab.py --workload c3 --k 31 --pack only "0,0,0" "1024,1,0" "2048,4,0" "4096,4,0" "512,1,0" "2048,1,0" > $OUT/ab_c3_packed.txt 2>$OUT/err.txt
python tools/ab.py --workload c3 --k 31 --pack only --u8 "0,0,0" "1024,1,0" > $OUT/ab_c3_packed_u8.txt 2>>$OUT/err.txt
python tools/ab.py --workload c3 --k 101 --pack only "0,0,0" "1024,1,0" "2048,4,0" > $OUT/ab_c3_k101_packed.txt 2>>$OUT/err.txt
python tools/ab.py --workload c5 --k 31 --pack only "0,0,0" "1024,1,0" "4096,4,0" > $OUT/ab_c5_packed.txt 2>>$OUT/err.txt
python tools/ab.py --workload c4 --k 31 --pack only "0,0,0" "512,4,2" "256,4,2" "1024,4,2" "512,1,2" > $OUT/ab_c4_packed.txt 2>>$OUT/err.txt
python tools/ab.py --workload c3 --k 31 --pack keep "0,0,0" > $OUT/ab_c3_keep.txt 2>>$OUT/err.txt
cat $OUT/ab_*.txt | cut -c1-220; grep -v "amdgpu.ids" $OUT/err.txt | tail -5
