#!/bin/bash
# round 3: the host packer alone on the GPU box's CPUs (no GPU in the loop): thread counts, SIMD on / off
TAG=${1:-r3hp}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{ nproc; grep -m1 "model name" /proc/cpuinfo; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; taskset -p $$; numactl -H 2>/dev/null | head -4; } > $OUT/hp.txt 2>&1
g++ -O3 -std=c++17 -pthread -I include tools/hostpack_bench.cpp memo_amd/csrc/memo_hostcore.cpp -o /tmp/hostpack_bench 2>> $OUT/hp.txt
for simd in 1 0; do for th in 8 16 32 64; do echo "MEMO_HOST_SIMD=$simd MEMO_HOST_THREADS=$th" >> $OUT/hp.txt
  MEMO_HOST_SIMD=$simd MEMO_HOST_THREADS=$th timeout 300 /tmp/hostpack_bench 200000000 1 >> $OUT/hp.txt 2>&1; done; done
cat $OUT/hp.txt
