#!/bin/bash
# round 6 evidence: bench lines (driver's command line, defaults), rocprofv3 kernel stats of that command, PMC traffic passes (separate
# runs; FETCH_SIZE / WRITE_SIZE) for every BASELINE config, the other workloads, the forced RCCL path at N = 1, the membership EXEC A/B
TAG=${1:-r6p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$GRAFT_REPO_ROOT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-160 $OUT/bench_driver.json
timeout 900 python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err; echo "bench default rc=$?"; cut -c1-160 $OUT/bench_default.json
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o c3 -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --headline-only > $ROOT/$OUT/bench_under_rocprof.json 2>> $ROOT/$OUT/prof.err )
head -8 $OUT/prof/c3_kernel_stats.csv | cut -c1-200
pmc() {  # key  kernel-substring  bench args...
  key=$1; kern=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp; timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$key/pmc_$c -o p -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --headline-only --calibrate "$@" > $ROOT/$OUT/bench_pmc_$key.json 2>> $ROOT/$OUT/prof.err )
  done
  python3 - <<PY
import json, os, subprocess
j = json.load(open("$OUT/bench_pmc_$key.json"))
env = dict(os.environ, ALG_BYTES=str(j["roofline"]["algorithmic_bytes"]), RESULT_BYTES=str(j["config"]["result_bytes_per_position"]))
r = subprocess.run(["python3", "tools/pmc_summary.py", "$key", "$OUT/pmc_$key", "$kern", "r06"], env=env, capture_output=True, text=True)
print("$key", "$kern", "alg", j["roofline"]["algorithmic_bytes"], (r.stdout[-260:] + r.stderr[-300:]).replace("\n", " "))
PY
}
pmc c3_dense sweep_conservation_halo3t_kernel
pmc c3_packed sweep_conservation_halo_kernel --rows packed
pmc c3_wide "sweep_conservation_kernel<" --rows wide
pmc c4_packed sweep_membership_planes_kernel --workload c4
pmc c5_dense sweep_conservation_halo3t_kernel --workload c5
pmc c5_packed_k101 sweep_conservation_mixed_kernel --workload c5 --k 101
pmc c5_dense_k21 sweep_conservation_halo3t_kernel --workload c5 --k 21
cp profiles/traffic.json $OUT/traffic.json
for wl in "c2 31" "c4 31" "c5 31" "c5 21" "c3 21" "c3 64" "c3 101" "c3 128" "c3 256" "c5 101" "c4 101" "c4 64"; do read -r w k <<< "$wl"
  timeout 400 python bench.py --workload $w --k $k --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.jsonl
done
python3 - <<PY
import json
for l in open("$OUT/workloads.jsonl"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:52], "k=%d"%j["config"]["k"], "| %.3g B rows: %.4f ms (median %.4f) frac %.3f val %.3g traffic %s"%(j["config"]["row_bytes"], r["kernel_ms"], r["kernel_ms_median"], r["frac"], j["value"], r["traffic"]), r["kernel"][:34])
PY
timeout 600 python bench.py --force-dist --launch --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_forced_rccl_one_rank.json 2>> $OUT/bench.err; cut -c1-200 $OUT/bench_forced_rccl_one_rank.json
if [ -f memo_amd/libmemo_amd_oldexec_ab.so ]; then
for rep in 1 2; do for lib in oldexec ab; do
  so=memo_amd/libmemo_amd_${lib}_ab.so; [ "$lib" = ab ] && so=memo_amd/libmemo_amd_ab.so
  for k in 31 21 101; do echo "== $lib c4 k=$k" >> $OUT/ab.txt; MEMO_AMD_AB_LIB=$so timeout 600 python tools/ab.py --workload c4 --k $k --pack only --prepare --rounds 40 1024,4,0 >> $OUT/ab.txt 2>> $OUT/ab.err; done
done; done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()[:150]); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
fi
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*agent_info*" -delete
grep -v "amdgpu.ids" $OUT/bench.err | tail -5; grep -v "^[EWI]2026" $OUT/prof.err | tail -5
