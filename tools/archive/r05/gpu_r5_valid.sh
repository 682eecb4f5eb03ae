#!/bin/bash
# round 5 validation: a long differential fuzz on the product build, then the sweeping tests and the fuzzer on the -DMEMO_EXEC_CHECK
# build of the A/B library (every branch-free row block -- the membership planes' since this round -- tests EXEC on entry), the bench line
TAG=${1:-r5valid}; SECS=${2:-240}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout $((SECS + 120)) python tests/fuzz_gpu.py --seconds $SECS > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-400
MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or views or level_arrays or row_order or random or tile or scatter or packed_k or 120 or prepare or cycling or memb or planes or bucket_widths" 2>&1 | tail -4 | tee $OUT/pytest_execcheck.txt | cut -c1-300
MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so timeout 300 python tests/fuzz_gpu.py --seconds 120 > $OUT/fuzz_execcheck.txt 2>&1; tail -2 $OUT/fuzz_execcheck.txt | cut -c1-400
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
j = json.load(open("$OUT/bench_driver.json"))
print("value %.4g  ms/step %.4f  frac %.3f traffic %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["traffic"]))
c = j["config"]
for kk in ("row_format_pass", "dense_format_pass", "dense_view_pass", "dense_view_place_pass"):
    print(kk, (c.get(kk) or {}).get("ms"))
PY
timeout 400 python bench.py --workload c4 --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err > $OUT/bench_c4.json; python -c "
import json; j=json.load(open('$OUT/bench_c4.json')); r=j['roofline']; print('c4', r['kernel_ms'], r['frac'], r['traffic'], j['value'])"
