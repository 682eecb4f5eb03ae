#!/bin/bash
# round 4: LDS counters of the headline kernel with and without the placing of a view's rows (tools/ab.py --no-colour): does the hardware
# count what tools/view_order_model.py predicts?
TAG=${1:-r4csq}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in colour nocolour; do
  fl=""; [ $v = nocolour ] && fl="--no-colour"
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
    d=$OUT/$v/$(echo $set | cut -c1-12 | tr ' ' '_'); mkdir -p $d
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o sq -- python tools/ab.py --workload c3 --k 31 --pack dense --u8 --rounds 30 $fl "0,0,0" > /dev/null 2>> $OUT/prof.err
    python - <<PY >> $OUT/sq.txt
import csv, glob, collections
f = glob.glob("$d/*counter_collection.csv")
rows = [r for r in csv.DictReader(open(f[0]))] if f else []
acc = collections.defaultdict(list)
for r in rows:
    if "halo3t" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    v = v[6:] if len(v) > 12 else v      # (the first launches read all the rows: the view is built by the fifth query)
    v.sort(); print("%-9s %-26s %.4g  (median of %d launches)" % ("$v", c, v[len(v) // 2], len(v)))
PY
  done
done
sort -k2,2 -k1,1 $OUT/sq.txt; tail -2 $OUT/prof.err
