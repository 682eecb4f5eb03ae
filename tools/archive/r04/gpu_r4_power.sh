#!/bin/bash
# round 4: the two states of the k >= 65 sweeps (profiles/r04_large_k.txt, items 4-5): power and clocks while one variant runs back to back
TAG=${1:-r4pw}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
probe() {  # label workload k variant
  python tools/ab.py --workload $2 --k $3 --pack only ${5} --rounds 25000 "$4" > $OUT/ab_$1.json 2>>$OUT/err.txt &
  PID=$!
  sleep 6
  for i in $(seq 1 60); do
    P=$(rocm-smi --showpower --showclocks 2>&1 | grep -iE "Power \(W\)|sclk|fclk|mclk" | sed -e 's/.*: //' | tr '\n' ' ')
    echo "$1 t=$i $P" >> $OUT/busy.txt
    kill -0 $PID 2>/dev/null || break
    sleep 1
  done
  wait $PID
  python -c "
import json
for l in open('$OUT/ab_$1.json'):
    j=json.loads(l); print('$1', 'median %.4f min %.4f max %.4f'%(j['ms_median'], j['ms_min'], j['ms_max']))" | tee -a $OUT/summary.txt
}
probe c5_k101_plan c5 101 "0,0,0"
probe c5_k101_all c5 101 "0,0,0,0,4"
probe c3_k101_plan c3 101 "0,0,0" --u8
probe c3_k101_all c3 101 "0,0,0,0,4" --u8
probe c5_k101_plan_again c5 101 "0,0,0"
awk '{print $1}' $OUT/busy.txt | sort | uniq -c; for l in c5_k101_plan c5_k101_all c3_k101_plan c3_k101_all c5_k101_plan_again; do grep "^$l " $OUT/busy.txt | sed -n '3p;8p'; done
