#!/bin/bash
TAG=${1:-sq}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+|GRBM_[A-Z_0-9]+)\b" | sort -u > $OUT/counters.txt; wc -l $OUT/counters.txt
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o c3 -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>> $OUT/err.txt
done
python - <<PY
import csv,glob,collections
for i in (1,2,3):
    fs=glob.glob("$OUT/p%d/*counter_collection.csv"%i)
    if not fs: print("pass",i,"no output"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "sweep_conservation" in r["Kernel_Name"]:
            kind="packed" if "PackedRows" in r["Kernel_Name"] else "wide"
            acc[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kind in acc:
        print(kind, {k: "%.4g"%(sum(v)/len(v)) for k,v in acc[kind].items()})
PY
grep -v "amdgpu.ids" $OUT/err.txt | grep -iE "error|invalid|not" | head -5
