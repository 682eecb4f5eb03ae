#!/bin/bash
# round 4, VERDICT r03 item 5: an index from sequences that cannot hide in the 256 MiB Infinity Cache: CHUNKS independent
# 20 Mbp x 50-genome pangenomes laid end to end on one pivot (tools/realistic_index.py --chunks), >= 1 GB of rows read per launch;
# bench lines at k = 21 / 31 / 101 + membership with whole-window parity, and FETCH_SIZE / WRITE_SIZE passes for k = 31 and 101
TAG=${1:-r4real}; CHUNKS=${2:-8}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
{ echo "memory.max $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"; free -g | head -2; df -h /tmp | tail -1; nproc; } > $OUT/box.txt 2>&1; cat $OUT/box.txt
LIM=$(cat /sys/fs/cgroup/memory.max 2>/dev/null); if [ "$LIM" != "max" ] && [ -n "$LIM" ] && [ "$LIM" -lt 120000000000 ]; then CHUNKS=4; echo "memory limit $LIM: $CHUNKS chunks" | tee -a $OUT/box.txt; fi
D=/tmp/real$CHUNKS
timeout 2400 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks $CHUNKS --out $D --threads 32 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"; cut -c1-400 $OUT/index_stats.json
ls -la $D | head; : > $OUT/bench.jsonl
L=$((CHUNKS * 20000000))
for k in 31 21 101; do
  timeout 900 python bench.py --rows-file $D/cons.npz --k $k --steps 200 --warmup 20 --cpu-sample $L >> $OUT/bench.jsonl 2>> $OUT/bench.err; echo "cons k=$k rc=$?"
done
timeout 900 python bench.py --rows-file $D/memb.npz --membership --k 31 --steps 100 --warmup 10 --cpu-sample 3000000 >> $OUT/bench.jsonl 2>> $OUT/bench.err; echo "memb rc=$?"
python - <<PY
import json
for l in open("$OUT/bench.jsonl"):
    j=json.loads(l); r=j["roofline"]; c=j["cpu_baseline"]
    print(j["config"]["query"], "k", j["config"]["k"], "rows", j["config"]["rows_per_gpu"], "read", j["config"]["rows_read"], "fmt", j["config"]["row_bytes"], r["kernel"][:44], "%.4f ms"%r["kernel_ms"], "%.3g pos/s"%j["value"], "frac %.3f"%r["frac"], "alg %.3g B" % r["algorithmic_bytes"],
          "cpu1 %.3g all %.3g"%(c["value"], c["all_cores"]["value"]), "parity", c["parity_with_gpu_on_sample"], [(o["rows"], round(o["kernel_ms"],4), round(o["frac"],3)) for o in j.get("other_row_formats", [])])
PY
# HBM traffic of the k = 31 and k = 101 launches: separate FETCH_SIZE / WRITE_SIZE passes (MI355X_MICROARCH.md), headline alone
for k in 31 101; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_k$k/pmc_$c -o real -- python bench.py --rows-file $D/cons.npz --k $k --steps 3 --warmup 1 --cpu-sample 0 --headline-only > $OUT/bench_pmc_k$k.json 2>> $OUT/prof.err
  done
  python - <<PY
import json, os, subprocess
j = json.load(open("$OUT/bench_pmc_k$k.json"))
kern = j["roofline"]["kernel"].split("<")[0]
key = "real${CHUNKS}_cons_k${k}_" + ("dense" if j["config"]["row_bytes"] == 3.2 else "packed")
env = dict(os.environ, ALG_BYTES=str(j["roofline"]["algorithmic_bytes"]), RESULT_BYTES=str(j["config"]["result_bytes_per_position"]))
print(key, kern, subprocess.run(["python", "tools/pmc_summary.py", key, "$OUT/pmc_k$k", kern, "r04"], env=env, capture_output=True, text=True).stdout[-900:])
PY
done
cp profiles/traffic.json $OUT/traffic.json
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*agent_info*" -delete
grep -v "amdgpu.ids" $OUT/bench.err | tail -5; tail -3 $OUT/prof.err
