#!/bin/bash
# diagnostic: what the conservation sweep would gain if locating a tile's rows cost no memory access (closed-form
# bucket table of the synthetic index, build -DMEMO_SYNTH_LOCATE); plus the transport tests again
TAG=${1:-r2n}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "transport" 2>&1 | tail -4 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
for rep in 1 2 3; do for lib in libmemo_amd_ab.so libmemo_amd_synthlocate_ab.so; do for k in 31 101; do
  printf "%-32s k=%-3s: " $lib $k >> $OUT/ab.txt
  MEMO_AMD_AB_LIB=$PWD/memo_amd/$lib timeout 300 python tools/ab.py --workload c3 --k $k --pack only --u8 --rounds 12 "0,0,0" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt; grep -v amdgpu.ids $OUT/err.txt | tail -5
timeout 300 python bench.py --force-dist --code-own-slice --nibble-gather --steps 20 --warmup 5 --cpu-sample 0 2>> $OUT/bench.err | cut -c1-300
