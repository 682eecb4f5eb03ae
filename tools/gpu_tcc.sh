#!/bin/bash
# L2 / memory-side counters of the bench workload's sweep kernels: gpu_tcc.sh tag [bench args]
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
i=0
for P in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum"; do i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o x -- python bench.py --steps 3 --warmup 1 --cpu-sample 0 "$@" > /dev/null 2>> $OUT/err.txt
done
python - <<PY
import csv,glob,collections
for i in (1,2,3):
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
grep -v "amdgpu.ids" $OUT/err.txt | grep -iE "error|invalid|not supported|unknown" | head -5
