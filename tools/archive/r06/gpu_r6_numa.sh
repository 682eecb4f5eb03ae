#!/bin/bash
# round 6: the NUMA A/B the verdict asked for -- workers pinned to the nodes in turn, every block packed by a worker of the node its
# pages lie on (libmemo_amd_numa.so: memo_hostcore.cpp built with -DMEMO_NUMA_EXPERIMENT) against the product (no pinning, blocks
# in order); columns first touched by the pool's threads (pages on both nodes) and by one thread (one node); alternating, two rounds
TAG=${1:-r6numa}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rep in 1 2; do
  echo "== product"; timeout 600 python tools/oneshot_sweep.py 4 2>&1 | grep -v amdgpu.ids | grep "touched\|rows form"
  echo "== numa";    MEMO_AMD_LIB=$PWD/memo_amd/libmemo_amd_numa.so timeout 600 python tools/oneshot_sweep.py 4 2>&1 | grep -v amdgpu.ids | grep "touched\|rows form"
done 2>&1 | tee $OUT/numa.txt
