#!/bin/bash
# SQ / store-path counters for one bench workload: gpu_sq2.sh tag workload [bench args]
TAG=$1; WL=$2; shift 2; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR"
P3="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_BRANCH"
P4="TCP_TCC_WRITE_REQ TCP_TOTAL_WRITE TCP_PENDING_STALL_CYCLES TCP_TOTAL_ACCESSES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o x -- python bench.py --workload $WL --steps 3 --warmup 1 --cpu-sample 0 "$@" > /dev/null 2>> $OUT/err.txt
done
python - <<PY
import csv,glob,collections
for i in (1,2,3,4):
    fs=glob.glob("$OUT/p%d/*counter_collection.csv"%i)
    if not fs: print("pass",i,"no output"); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if "sweep_" in r["Kernel_Name"]:
            nm=r["Kernel_Name"]; kind=("packed " if "PackedRows" in nm else "wide ")+nm.split("sweep_")[1].split("<")[0]
            acc[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kind in sorted(acc):
        for k,v in sorted(acc[kind].items()): print(f"{kind:32s} {k:30s} {sum(v)/len(v):.4g}")
PY
grep -v "amdgpu.ids" $OUT/err.txt | grep -iE "error|invalid" | head -5
