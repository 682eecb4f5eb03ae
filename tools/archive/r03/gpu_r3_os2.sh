#!/bin/bash
# round 3: one-shot seam, the host packer before (row at a time, 65 536-row tasks: libmemo_amd_oldpack.so = this build's objects
# with HEAD~'s memo_hostcore.cpp) and after (320-row vectorised pieces, 16 384-row tasks), alternating on one box
TAG=${1:-r3os2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc $(nproc)" > $OUT/ab.txt
for rep in 1 2 3; do
  for lib in memo_amd/libmemo_amd_oldpack.so memo_amd/libmemo_amd.so; do
    echo "== $lib" >> $OUT/ab.txt
    MEMO_AMD_LIB=$PWD/$lib timeout 600 python tools/oneshot_timing.py --big-only 2>&1 | grep -v amdgpu.ids >> $OUT/ab.txt
  done
done
cat $OUT/ab.txt
