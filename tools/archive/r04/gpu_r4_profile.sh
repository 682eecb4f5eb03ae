#!/bin/bash
# round 4 evidence: smoke, full GPU tier, bench (driver command line + defaults), rocprofv3 kernel stats of that command, PMC traffic
# passes (separate runs), the other workloads, SQ counters of the k >= 65 kernels, the forced RCCL path at N = 1, fuzz
TAG=${1:-r4p}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
[ "$2" = "nopytest" ] || { timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt | cut -c1-300; }
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench_driver.json
timeout 600 python bench.py > $OUT/bench_default.json 2>> $OUT/bench.err; echo "bench default rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c3 -- python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --headline-only > $OUT/bench_under_rocprof.json 2>> $OUT/prof.err
head -12 $OUT/prof/c3_kernel_stats.csv | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do   # pass A: the headline alone
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcA/pmc_$c -o c3 -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --headline-only > $OUT/bench_pmc.json 2>> $OUT/prof.err
done
ALG=$(python -c "import json; print(json.load(open('$OUT/bench_pmc.json'))['roofline']['algorithmic_bytes'])")
ALG_BYTES=$ALG python tools/pmc_summary.py c3_dense $OUT/pmcA "sweep_conservation_halo3t_kernel" r04 > $OUT/traffic_dense.txt 2>&1; tail -8 $OUT/traffic_dense.txt
for c in FETCH_SIZE WRITE_SIZE; do   # pass B: the other resident formats + the calibration kernel
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmcB/pmc_$c -o c3 -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 --calibrate > /dev/null 2>> $OUT/prof.err
done
python tools/pmc_summary.py c3_packed $OUT/pmcB "sweep_conservation_halo_kernel" r04 > $OUT/traffic_packed.txt 2>&1
python tools/pmc_summary.py c3_wide $OUT/pmcB "sweep_conservation_kernel" r04 > $OUT/traffic_wide.txt 2>&1
cp profiles/traffic.json $OUT/traffic.json
for wl in "c2 31" "c4 31" "c5 31" "c5 21" "c3 21" "c3 64" "c3 101" "c3 128" "c3 256" "c5 101" "c4 101"; do read -r w k <<< "$wl"
  timeout 400 python bench.py --workload $w --k $k --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.jsonl
done
python - <<PY
import json
for l in open("$OUT/workloads.jsonl"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:52], "k=%d"%j["config"]["k"], "| %s B rows: %.3f ms (median %.3f) frac %.3f val %.3g"%(j["config"]["row_bytes"], r["kernel_ms"], r["kernel_ms_median"], r["frac"], j["value"]), r["kernel"][:34],
          "| others:", ["%s %.3f ms frac %.3f"%(o["rows"][:12], o["kernel_ms_median"], o["frac"]) for o in j.get("other_row_formats", [])])
PY
# SQ / LDS counters of the k >= 65 kernels (SQ_INSTS_SALU 1.64e8, conflict ratio 0.51, SQ_WAIT_INST_LDS 2.1e8 at k = 101)
for spec in "c3 101" "c3 256" "c5 101"; do read -r w k <<< "$spec"
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD"; do
    d=$OUT/sq_${w}_k$k/$(echo $set | cut -c1-12 | tr ' ' '_'); mkdir -p $d
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o sq -- python bench.py --workload $w --k $k --steps 3 --warmup 1 --cpu-sample 0 --headline-only > /dev/null 2>> $OUT/prof.err
    python - <<PY >> $OUT/sq_counters.txt
import csv, glob, collections
f = glob.glob("$d/*counter_collection.csv")
rows = [r for r in csv.DictReader(open(f[0]))] if f else []
acc = collections.defaultdict(list)
for r in rows:
    if "sweep_" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"].split("<")[0].replace("void (anonymous namespace)::", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kn, c), v in sorted(acc.items()):
    v.sort(); print("$w k=$k %-40s %-26s %.4g  (median of %d launches)" % (kn, c, v[len(v) // 2], len(v)))
PY
  done
done
cat $OUT/sq_counters.txt | cut -c1-150
timeout 600 python bench.py --gpus 1 --launch --force-dist --steps 20 --warmup 5 --cpu-sample 0 > $OUT/bench_dist.json 2> $OUT/bench_dist.err; echo "dist rc=$?"
python - <<PY
import json
a=json.load(open("$OUT/bench_driver.json")); b=json.load(open("$OUT/bench_dist.json"))
print("forced RCCL path at N=1: value %.4g vs plain %.4g: %+.2f %%"%(b["value"], a["value"], 100*(b["value"]/a["value"]-1)), b.get("ranks_seen",{}).get("distinct_devices"), b["config"].get("gather_payload"))
PY
timeout 500 python tests/fuzz_gpu.py --seconds ${3:-360} > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-500
find $OUT -name "*.csv" -size +3M -delete; find $OUT -name "*agent_info*" -delete
grep -v "amdgpu.ids\|socket.cpp" $OUT/bench.err | tail -5
