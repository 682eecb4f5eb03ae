#!/bin/bash
# clocks and power while the config-3 sweep runs back to back (is the steady state a power-capped one?)
TAG=${1:-pw}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for rows in only dense; do
  python tools/ab.py --workload c3 --k 31 --pack $rows --u8 --rounds 40000 "0,0,0" > $OUT/ab_$rows.json 2>>$OUT/err.txt &
  PID=$!
  for i in $(seq 1 45); do
    P=$(rocm-smi --showpower --showclocks 2>&1 | grep -iE "Power \(W\)|sclk|fclk" | sed -e 's/.*: //' | tr '\n' ' ')
    echo "$rows t=$i $P" >> $OUT/busy.txt
    kill -0 $PID 2>/dev/null || break
    sleep 1
  done
  wait $PID
  python -c "
import json,sys
for l in open('$OUT/ab_$rows.json'):
    j=json.loads(l); print('$rows', 'median %.4f min %.4f'%(j['ms_median'], j['ms_min']))"
done
cat $OUT/busy.txt
