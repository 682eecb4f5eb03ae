#!/bin/bash
# round 6: the sequence-built index that cannot hide in the Infinity Cache (160 Mbp x 50 genomes: tools/realistic_index.py --chunks 8):
# bench lines at k = 31 / 21 / 101 + membership with whole-window parity (the table-driven kernel now picks its SP form for it)
TAG=${1:-r6real}; CHUNKS=${2:-8}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
LIM=$(cat /sys/fs/cgroup/memory.max 2>/dev/null); if [ "$LIM" != "max" ] && [ -n "$LIM" ] && [ "$LIM" -lt 120000000000 ]; then CHUNKS=4; echo "memory limit $LIM: $CHUNKS chunks" | tee $OUT/box.txt; fi
D=/tmp/real$CHUNKS
timeout 2400 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks $CHUNKS --out $D --threads 16 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"
: > $OUT/bench.jsonl
L=$((CHUNKS * 20000000))
for k in 31 21 101; do
  timeout 900 python bench.py --rows-file $D/cons.npz --k $k --steps 200 --warmup 20 --cpu-sample $L >> $OUT/bench.jsonl 2>> $OUT/bench.err; echo "cons k=$k rc=$?"
done
timeout 900 python bench.py --rows-file $D/memb.npz --membership --k 31 --steps 100 --warmup 10 --cpu-sample 3000000 >> $OUT/bench.jsonl 2>> $OUT/bench.err; echo "memb rc=$?"
python - <<PY
import json
for l in open("$OUT/bench.jsonl"):
    j=json.loads(l); r=j["roofline"]; c=j["cpu_baseline"]
    print(j["config"]["query"], "k", j["config"]["k"], "rows", j["config"]["rows_per_gpu"], "read", j["config"]["rows_read"], "fmt %.3g" % j["config"]["row_bytes"], r["kernel"][:44], "%.4f ms"%r["kernel_ms"], "%.3g pos/s"%j["value"], "frac %.3f"%r["frac"], "alg %.3g B" % r["algorithmic_bytes"],
          "parity", c["parity_with_gpu_on_sample"], [(o["rows"][:28], round(o["kernel_ms"],4), round(o["frac"],3)) for o in j.get("other_row_formats", [])])
PY
grep -v "amdgpu.ids" $OUT/bench.err | tail -5
