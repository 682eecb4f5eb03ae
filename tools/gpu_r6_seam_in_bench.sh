#!/bin/bash
# why is the seam slower inside bench.py than in tools/oneshot_sweep.py on some boxes?  the library's phase lines of both, the box's state
TAG=${1:-r6seam}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
echo "thp $(cat /sys/kernel/mm/transparent_hugepage/enabled)  defrag $(cat /sys/kernel/mm/transparent_hugepage/defrag)  loadavg $(cat /proc/loadavg)  cpu.max $(cat /sys/fs/cgroup/cpu.max)"
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
MEMO_TIMING=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
grep "memo one-shot" $OUT/bench.err | sed 's/memo one-shot: 499999995 rows packed to 3.2 B (dense rows): //'
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
echo "== standalone"
timeout 600 python tools/oneshot_sweep.py 3 2>&1 | grep -v amdgpu.ids | sed 's/memo one-shot: 499999995 rows packed to 3.2 B (dense rows): //'
grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
