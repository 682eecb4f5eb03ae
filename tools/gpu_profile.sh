#!/bin/bash
# GPU box: bench line + rocprofv3 kernel stats + PMC passes (separate runs) for both row formats, then the
# other workloads; sparse-index A/B of the two scatters.
TAG=${1:-r05}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
for rows in packed wide; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$rows -o c3 -- python bench.py --rows $rows --steps 100 --warmup 20 --cpu-sample 0 > $OUT/prof_$rows.json 2>> $OUT/prof.err
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$rows/pmc_$c -o c3 -- python bench.py --rows $rows --steps 3 --warmup 1 --cpu-sample 0 --calibrate > /dev/null 2>> $OUT/prof.err
  done
done
head -6 $OUT/prof_packed/c3_kernel_stats.csv | cut -c1-200
for wl in c2 c4 c5; do timeout 300 python bench.py --workload $wl --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.txt; done
timeout 300 python bench.py --k 101 --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.txt
timeout 300 python bench.py --k 21 --steps 100 --warmup 20 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.txt
timeout 300 python bench.py --force-dist --steps 10 --warmup 2 --cpu-sample 0 2>>$OUT/bench.err >> $OUT/workloads.txt
python - <<PY
import json
for l in open("$OUT/workloads.txt"):
    j=json.loads(l); r=j["roofline"]; o=j.get("other_row_format") or {}
    print(j["config"]["workload"][:52], "k=%d"%j["config"]["k"], "| %s: %.3f ms frac %.3f val %.3g | other: %s %.3f ms frac %.3f"%(j["config"]["row_bytes"], r["kernel_ms"], r["frac"], j["value"], o.get("rows"), o.get("kernel_ms",0), o.get("frac",0)), j.get("gather_parity_sample",""))
PY
for wl in "sparse 31" "c2 31"; do read -r w k <<< "$wl"
  echo "== $w k=$k packed u8" >> $OUT/ab.txt
  python tools/ab.py --workload $w --k $k --pack only --u8 --rounds 10 "0,0,0,0,1" "0,0,0,0,2" 2>>$OUT/err.txt >> $OUT/ab.txt
done
cat $OUT/ab.txt
grep -v "amdgpu.ids\|socket.cpp" $OUT/bench.err | tail -5
