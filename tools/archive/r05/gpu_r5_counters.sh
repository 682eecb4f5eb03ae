#!/bin/bash
# round 5: hardware counters (separate rocprofv3 --pmc passes, kernel trace only) of
#   (a) the table-driven conservation sweep on five- against six-row views at k = 9 / 17 / 25 / 31 (the six-row
#       views' k = 17 gain is larger than their bytes explain) -> $OUT/six_rows.txt
#   (b) with REAL=dir: the sequence-built index's k = 101 conservation and k = 31 membership sweeps (item 2) -> $OUT/sq_realistic.txt
TAG=${1:-r5cnt}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$GRAFT_REPO_ROOT
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"
      "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum")
summ() {  # dir label pattern
python3 - "$1" "$2" "$3" <<'PY'
import csv, glob, collections, sys
d, label, pat = sys.argv[1:4]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    v.sort()
    print("%-44s %-28s %.5g  (median of %d launches)" % (label, c, v[len(v) // 2], len(v)))
PY
}
cd /tmp
if [ -z "$REAL" ]; then
  for k in 9 17 25 31; do for rpg in 5 6; do
    i=0; for set in "${SETS[@]}"; do i=$((i+1)); d=$ROOT/$OUT/six/k${k}_r${rpg}_s$i; mkdir -p $d
      timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o sq -- python3 $ROOT/tools/view_sweep_ab.py --ks $k --variants 0:$rpg:1 --reps 1 --launches 40 > /dev/null 2>> $ROOT/$OUT/prof.err
    done
    summ $ROOT/$OUT/six/k${k}_r${rpg}_s1 "k=$k rows_per_group=$rpg" sweep_conservation_halo3t >> $ROOT/$OUT/six_rows.txt
    summ $ROOT/$OUT/six/k${k}_r${rpg}_s2 "k=$k rows_per_group=$rpg" sweep_conservation_halo3t >> $ROOT/$OUT/six_rows.txt
    summ $ROOT/$OUT/six/k${k}_r${rpg}_s3 "k=$k rows_per_group=$rpg" sweep_conservation_halo3t >> $ROOT/$OUT/six_rows.txt
    timeout 200 python3 $ROOT/tools/view_sweep_ab.py --ks $k --variants 0:$rpg:1 --reps 2 --launches 900 2>/dev/null >> $ROOT/$OUT/six_rows_times.jsonl
  done; done
  cat $ROOT/$OUT/six_rows.txt | cut -c1-140
else
  for spec in "cons 101" "memb 31"; do read -r q k <<< "$spec"
    extra=""; [ $q = memb ] && extra="--membership"
    i=0; for set in "${SETS[@]}"; do i=$((i+1)); d=$ROOT/$OUT/real/${q}_k${k}_s$i; mkdir -p $d
      timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o sq -- python3 $ROOT/bench.py --rows-file $REAL/$q.npz $extra --k $k --steps 3 --warmup 1 --cpu-sample 0 --headline-only > /dev/null 2>> $ROOT/$OUT/prof.err
      summ $d "realistic $q k=$k" sweep_ >> $ROOT/$OUT/sq_realistic.txt
    done
  done
  cat $ROOT/$OUT/sq_realistic.txt | cut -c1-150
fi
find $ROOT/$OUT -name "*.csv" -size +1M -delete; find $ROOT/$OUT -name "*agent_info*" -delete
