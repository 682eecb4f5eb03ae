#!/bin/bash
# round 5 evidence, second half: PMC traffic passes (separate runs; FETCH_SIZE / WRITE_SIZE, kernel trace only) for every BASELINE
# config -> profiles/traffic.json, THEN the bench lines (so that roofline.traffic is filled from this box's passes) and the kernel stats
TAG=${1:-r5pmc}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$GRAFT_REPO_ROOT
pmc() {  # key  kernel-substring  bench args...
  key=$1; kern=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$key/pmc_$c -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --headline-only --calibrate "$@" > $ROOT/$OUT/bench_pmc_$key.json 2>> $ROOT/$OUT/prof.err )
  done
  python3 - <<PY
import json, os, subprocess
j = json.load(open("$OUT/bench_pmc_$key.json"))
env = dict(os.environ, ALG_BYTES=str(j["roofline"]["algorithmic_bytes"]), RESULT_BYTES=str(j["config"]["result_bytes_per_position"]))
r = subprocess.run(["python3", "tools/pmc_summary.py", "$key", "$OUT/pmc_$key", "$kern", "r05"], env=env, capture_output=True, text=True)
print("$key", "$kern", "alg", j["roofline"]["algorithmic_bytes"], (r.stdout[-260:] + r.stderr[-300:]).replace("\n", " "))
PY
}
pmc c3_dense sweep_conservation_halo3t_kernel
pmc c3_packed sweep_conservation_halo_kernel --rows packed
pmc c4_packed sweep_membership_planes_kernel --workload c4
pmc c5_dense sweep_conservation_halo3t_kernel --workload c5
pmc c5_packed_k101 sweep_conservation_mixed_kernel --workload c5 --k 101
pmc c5_dense_k21 sweep_conservation_halo3t_kernel --workload c5 --k 21
cp profiles/traffic.json $OUT/traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-160 $OUT/bench_driver.json
timeout 900 python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err; echo "bench default rc=$?"; cut -c1-160 $OUT/bench_default.json
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o c3 -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --headline-only > $ROOT/$OUT/bench_under_rocprof.json 2>> $ROOT/$OUT/prof.err )
head -4 $OUT/prof/c3_kernel_stats.csv | cut -c1-200
for wl in "c4 31" "c5 31" "c5 21" "c5 101" "c4 101"; do read -r w k <<< "$wl"
  timeout 400 python bench.py --workload $w --k $k --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.jsonl
done
python3 - <<PY
import json
for l in open("$OUT/workloads.jsonl"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:52], "k=%d"%j["config"]["k"], "| %.3g B rows: %.4f ms (median %.4f) frac %.3f val %.3g traffic %s"%(j["config"]["row_bytes"], r["kernel_ms"], r["kernel_ms_median"], r["frac"], j["value"], r["traffic"]), r["kernel"][:34])
PY
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*agent_info*" -delete
grep -v "amdgpu.ids" $OUT/bench.err | tail -5; grep -v "^[EWI]2026" $OUT/prof.err | tail -5
