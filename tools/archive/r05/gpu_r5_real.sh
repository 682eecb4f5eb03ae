#!/bin/bash
# round 5, VERDICT r04 item 2: the sequence-built index that cannot hide in the Infinity Cache (160 Mbp x 50 genomes: 8 independent
# 20 Mbp pangenomes end to end, tools/realistic_index.py --chunks 8): bench lines at k = 31 / 21 / 101 + membership with whole-window
# parity, then the SQ / LDS counters of the k = 101 and membership sweeps (tools/gpu_r5_counters.sh with REAL=...)
TAG=${1:-r5real}; CHUNKS=${2:-8}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
LIM=$(cat /sys/fs/cgroup/memory.max 2>/dev/null); if [ "$LIM" != "max" ] && [ -n "$LIM" ] && [ "$LIM" -lt 120000000000 ]; then CHUNKS=4; echo "memory limit $LIM: $CHUNKS chunks" | tee $OUT/box.txt; fi
D=/tmp/real$CHUNKS
timeout 2400 python tools/realistic_index.py --length 20000000 --genomes 50 --chunks $CHUNKS --out $D --threads 32 > $OUT/index_stats.json 2> $OUT/index.err; echo "index rc=$?"
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
# PMC traffic of the same lines (separate runs per counter, kernel trace only): profiles/traffic.json keys real<chunks>_{cons,memb}_k<k>_<rows>
ROOT=$GRAFT_REPO_ROOT
pmc() {  # key  kernel-substring  bench args...
  key=$1; kern=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$key/pmc_$c -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --headline-only "$@" > $ROOT/$OUT/bench_pmc_$key.json 2>> $ROOT/$OUT/prof.err )
  done
  python3 - <<PY
import json, os, subprocess
j = json.load(open("$OUT/bench_pmc_$key.json"))
env = dict(os.environ, ALG_BYTES=str(j["roofline"]["algorithmic_bytes"]), RESULT_BYTES=str(j["config"]["result_bytes_per_position"]))
r = subprocess.run(["python3", "tools/pmc_summary.py", "$key", "$OUT/pmc_$key", "$kern", "r05"], env=env, capture_output=True, text=True)
print("$key", "$kern", "alg", j["roofline"]["algorithmic_bytes"], (r.stdout[-200:] + r.stderr[-300:]).replace("\n", " "))
PY
}
if [ -z "$NO_PMC" ]; then
  pmc real${CHUNKS}_cons_k31_dense sweep_conservation_halo3t_kernel --rows-file $D/cons.npz --k 31
  pmc real${CHUNKS}_cons_k101_packed sweep_conservation_r4_kernel --rows-file $D/cons.npz --k 101
  pmc real${CHUNKS}_memb_k31_packed sweep_membership_planes_kernel --rows-file $D/memb.npz --membership --k 31
  cp profiles/traffic.json $OUT/traffic.json
fi
[ -z "$NO_SQ" ] && REAL=$D bash tools/gpu_r5_counters.sh $TAG
grep -v "amdgpu.ids" $OUT/bench.err | tail -5
