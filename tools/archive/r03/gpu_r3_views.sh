#!/bin/bash
TAG=${1:-r3v}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu -k "dense or sidecar or realistic or golden_one_shot or goldens_on or config3_full_size" 2>&1 | tail -8 | cut -c1-400
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
j=json.load(open("$OUT/bench_driver.json")); r=j["roofline"]
print("value %.4g  ms/step %.4f  kernel_ms %.4f  frac %.3f  rows_read %s  view %s  parity %s" % (j["value"], j["ms_per_step"], r["kernel_ms"], r["frac"], j["config"].get("rows_read"), j["config"].get("dense_view_pass"), j["cpu_baseline"]["parity_with_gpu_on_sample"]))
print(r["kernel"]); print([(o["rows"], round(o["kernel_ms"],4), round(o["frac"],3)) for o in j.get("other_row_formats",[])])
PY
for rep in 1 2; do for v in "0,0,0,9" "0,0,0,0"; do for k in 31 21 64; do
  printf "c3 k=%-3s %-10s: " $k $v >> $OUT/ab.txt
  timeout 300 python tools/ab.py --workload c3 --k $k --pack dense --u8 --rounds 2000 "$v" 2>>$OUT/err.txt | python -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('%.4f ms median  min %.4f'%(j['ms_median'], j['ms_min']))" >> $OUT/ab.txt
done; done; done
sort $OUT/ab.txt
