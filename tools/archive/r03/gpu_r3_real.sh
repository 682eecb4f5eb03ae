#!/bin/bash
# round 3, VERDICT item 6: an index from sequences, measured.  50 genomes x 5 Mbp by default.
TAG=${1:-r3real}; LEN=${2:-5000000}; N=${3:-50}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python tools/realistic_index.py --length $LEN --genomes $N --out /tmp/real --threads ${4:-32} > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"; cut -c1-600 $OUT/index_stats.json
: > $OUT/bench.jsonl
for k in 21 31 101; do
  timeout 600 python bench.py --rows-file /tmp/real/cons.npz --k $k --steps 200 --warmup 20 --cpu-sample $LEN >> $OUT/bench.jsonl 2>> $OUT/bench.err; echo "cons k=$k rc=$?"
done
timeout 600 python bench.py --rows-file /tmp/real/memb.npz --membership --k 31 --steps 100 --warmup 10 --cpu-sample 3000000 >> $OUT/bench.jsonl 2>> $OUT/bench.err; echo "memb rc=$?"
python - <<PY
import json
for l in open("$OUT/bench.jsonl"):
    j=json.loads(l); r=j["roofline"]; c=j["cpu_baseline"]
    print(j["config"]["query"], "k", j["config"]["k"], "rows", j["config"]["rows_per_gpu"], "fmt", j["config"]["row_bytes"], r["kernel"][:44], "%.4f ms"%r["kernel_ms"], "%.3g pos/s"%j["value"], "frac %.3f"%r["frac"],
          "cpu1 %.3g all %.3g"%(c["value"], c["all_cores"]["value"]), "parity", c["parity_with_gpu_on_sample"], [(o["rows"], round(o["kernel_ms"],4), o["kernel"][:30]) for o in j.get("other_row_formats", [])])
PY
grep -v "amdgpu.ids" $OUT/bench.err | tail -5
