#!/bin/bash
TAG=${1:-ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python tools/ab.py --workload c3 --k 31 "1024,1,0" "512,1,0" "2048,4,0" "4096,4,0" "1024,4,0" > $OUT/ab_c3_k31.txt 2>$OUT/err.txt
python tools/ab.py --workload c3 --k 101 "1024,1,0" "512,1,0" "2048,4,0" "4096,4,0" > $OUT/ab_c3_k101.txt 2>>$OUT/err.txt
python tools/ab.py --workload c3 --k 21 "1024,1,0" "4096,4,0" > $OUT/ab_c3_k21.txt 2>>$OUT/err.txt
python tools/ab.py --workload c5 --k 31 "1024,1,0" "4096,4,0" "2048,4,0" > $OUT/ab_c5.txt 2>>$OUT/err.txt
python tools/ab.py --workload c4 --k 31 "512,4,2" "256,4,2" "512,1,2" "1024,4,2" "512,1,1" "2048,4,1" > $OUT/ab_c4.txt 2>>$OUT/err.txt
python tools/ab.py --workload c2 --k 31 "1024,1,0" "4096,4,0" "512,1,0" "256,1,0" > $OUT/ab_c2.txt 2>>$OUT/err.txt
cat $OUT/ab_*.txt; grep -v "amdgpu.ids" $OUT/err.txt | tail -5
