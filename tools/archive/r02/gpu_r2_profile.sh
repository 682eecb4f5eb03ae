#!/bin/bash
# round 2 evidence: bench (driver command line), rocprofv3 kernel stats of that command, PMC traffic passes
# (separate runs), SQ counters, the other workloads, CLI phases with and without the sidecar cache
TAG=${1:-r2p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_driver.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c3 -- python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_under_rocprof.json 2>> $OUT/prof.err
head -12 $OUT/prof/c3_kernel_stats.csv | cut -c1-220
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc/pmc_$c -o c3 -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --calibrate > /dev/null 2>> $OUT/prof.err
done
python tools/pmc_summary.py c3_packed $OUT/pmc "sweep_conservation_halo_kernel" r02 > $OUT/traffic_packed.txt 2>&1; tail -12 $OUT/traffic_packed.txt
python tools/pmc_summary.py c3_wide $OUT/pmc "sweep_conservation_kernel" r02 > $OUT/traffic_wide.txt 2>&1
python tools/pmc_summary.py c3_dense $OUT/pmc "sweep_conservation_halo3_kernel" r02 > $OUT/traffic_dense.txt 2>&1
cp profiles/traffic.json $OUT/traffic.json
for wl in "c2 31" "c4 31" "c5 31" "c3 21" "c3 101" "c3 256" "c5 101" "c4 101"; do read -r w k <<< "$wl"
  timeout 400 python bench.py --workload $w --k $k --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.jsonl
done
python - <<PY
import json
for l in open("$OUT/workloads.jsonl"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:52], "k=%d"%j["config"]["k"], "| %s B rows: %.3f ms (median %.3f) frac %.3f val %.3g"%(j["config"]["row_bytes"], r["kernel_ms"], r["kernel_ms_median"], r["frac"], j["value"]),
          "| others:", ["%s %.3f ms frac %.3f"%(o["rows"], o["kernel_ms_median"], o["frac"]) for o in j.get("other_row_formats", [])])
PY
bash tools/gpu_sq.sh $TAG/sq31 c3 > $OUT/sq_counters_k31.txt 2>&1; bash tools/gpu_sq.sh $TAG/sq101 c3 --k 101 > $OUT/sq_counters_k101.txt 2>&1
grep "LDS_BANK\|LDS_IDX\|INSTS_VALU\|INSTS_LDS\|WAVE_CYCLES" $OUT/sq_counters_k31.txt | head -12
nproc > $OUT/cli_timing.txt
timeout 900 python tools/cli_timing.py --num-docs 100 --pivot 20000000 --out /tmp/cli_t >> $OUT/cli_timing.txt 2>&1; cat $OUT/cli_timing.txt
grep -v "amdgpu.ids\|socket.cpp" $OUT/bench.err | tail -5
