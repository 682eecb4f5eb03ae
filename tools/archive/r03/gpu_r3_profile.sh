#!/bin/bash
# round 3 evidence: full GPU tier, bench (driver command line), rocprofv3 kernel stats of that command, PMC traffic passes
# (separate runs), SQ counters, the other workloads, the forced RCCL path at N = 1
TAG=${1:-r3p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
[ "$2" = "nopytest" ] || { timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt | cut -c1-300; }
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench_driver.json
timeout 600 python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err; echo "bench default rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c3 -- python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --headline-only > $OUT/bench_under_rocprof.json 2>> $OUT/prof.err
head -14 $OUT/prof/c3_kernel_stats.csv | cut -c1-220
for c in FETCH_SIZE WRITE_SIZE; do   # pass A: the headline alone (every dense-row launch but the first four reads the k-class view)
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcA/pmc_$c -o c3 -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --headline-only > $OUT/bench_pmc.json 2>> $OUT/prof.err
done
ALG=$(python -c "import json; print(json.load(open('$OUT/bench_pmc.json'))['roofline']['algorithmic_bytes'])")
ALG_BYTES=$ALG python tools/pmc_summary.py c3_dense $OUT/pmcA "sweep_conservation_halo3t_kernel" r03 > $OUT/traffic_dense.txt 2>&1; tail -14 $OUT/traffic_dense.txt
for c in FETCH_SIZE WRITE_SIZE; do   # pass B: the other resident formats + the calibration kernel
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcB/pmc_$c -o c3 -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --calibrate > /dev/null 2>> $OUT/prof.err
done
python tools/pmc_summary.py c3_packed $OUT/pmcB "sweep_conservation_halo_kernel" r03 > $OUT/traffic_packed.txt 2>&1
python tools/pmc_summary.py c3_wide $OUT/pmcB "sweep_conservation_kernel" r03 > $OUT/traffic_wide.txt 2>&1
cp profiles/traffic.json $OUT/traffic.json
for wl in "c2 31" "c4 31" "c5 31" "c3 21" "c3 64" "c3 101" "c3 256" "c5 101" "c4 101"; do read -r w k <<< "$wl"
  timeout 400 python bench.py --workload $w --k $k --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.jsonl
done
python - <<PY
import json
for l in open("$OUT/workloads.jsonl"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:52], "k=%d"%j["config"]["k"], "| %s B rows: %.3f ms (median %.3f) frac %.3f val %.3g"%(j["config"]["row_bytes"], r["kernel_ms"], r["kernel_ms_median"], r["frac"], j["value"]),
          "| others:", ["%s %.3f ms frac %.3f"%(o["rows"][:12], o["kernel_ms_median"], o["frac"]) for o in j.get("other_row_formats", [])])
PY
bash tools/gpu_sq.sh $TAG/sq31 c3 > $OUT/sq_counters_k31.txt 2>&1
grep "LDS_BANK\|LDS_IDX\|INSTS_VALU\|INSTS_SALU\|INSTS_LDS\|WAVE_CYCLES\|BUSY_CYCLES" $OUT/sq_counters_k31.txt | head -24
timeout 600 python bench.py --gpus 1 --launch --force-dist --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_dist.json 2> $OUT/bench_dist.err; echo "dist rc=$?"
python - <<PY
import json
a=json.load(open("$OUT/bench_driver.json")); b=json.load(open("$OUT/bench_dist.json"))
print("forced RCCL path at N=1: value %.4g vs plain %.4g: %+.2f %%"%(b["value"], a["value"], 100*(b["value"]/a["value"]-1)), b.get("ranks_seen",{}).get("distinct_devices"), b["config"].get("gather_payload"))
PY
grep -v "amdgpu.ids\|socket.cpp" $OUT/bench.err | tail -5
