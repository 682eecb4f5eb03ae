#!/bin/bash
# the GPU box's host: topology, cgroup quota, read bandwidth by thread count, the packer alone by thread count
TAG=${1:-hostprobe}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
{
lscpu | grep -E 'Model name|Socket|Core|Thread|NUMA|^CPU\(s\)|L3|MHz'
free -g | head -2
g++ -O3 -mavx2 -std=c++17 -pthread tools/host_probe.cpp -o /tmp/host_probe && timeout 600 /tmp/host_probe 300000000
g++ -O3 -std=c++17 -pthread -I include -I memo_amd/csrc tools/hostpack_bench.cpp memo_amd/csrc/memo_hostcore.cpp -o /tmp/hostpack_bench
for t in 8 16 32 48 64 96 128; do MEMO_HOST_THREADS=$t timeout 300 /tmp/hostpack_bench 300000000 1 | tail -2; done
} > $OUT/host_probe.txt 2>&1
tail -100 $OUT/host_probe.txt
