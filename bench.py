#!/usr/bin/env python3
"""bench.py -- query-positions/sec of the windowed k-mer conservation sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json config 3 by default): synthetic pangenome, 100 genomes, 5 rows per
pivot position (density 0.05 per genome-position), one window of 10^8 positions per GPU,
k = 31, conservation.  A "step" is one query of the whole window: one launch of
sweep_conservation_kernel over the rows resident in HBM, result left in HBM.  Weak scaling:
every rank owns its own 10^8-position window of an N x 10^8 pivot and generates exactly the
rows that window sees (index-addressable generator, memo_amd/synth.py); for N > 1 the result
slices are gathered to rank 0 over RCCL inside the timed region.

One JSON line on stdout (rank 0).  `roofline` prices the sweep kernel alone from HIP events
recorded on the launch stream; `cpu_baseline` is the oracle's literal port of the
reference loop (memo_query.py:45-63,70) on one host core over a bounded sample window.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import memo_amd  # noqa: E402
from memo_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)

WORKLOADS = {
    # name: (num_docs, window positions per GPU, membership)
    "c2": (10, 10_000_000, False),
    "c3": (100, 100_000_000, False),
    "c4": (100, 100_000_000, True),
    "c5": (500, 1 << 25, False),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--cpu-sample", type=int, default=100_000_000,
                    help="positions of the window the 1-core CPU baseline is timed on (0 = skip)")
    return ap.parse_args()


def cpu_baseline(args, num_docs, L, k, membership, gpu_result_slice):
    """The oracle's literal transcription of memo_init + memo_query + argmax, one core,
    on the first `cpu_sample` positions of rank 0's window.  Also serves as the parity check
    of the benchmarked result (same positions)."""
    from oracle import memo_oracle as oracle        # checker / baseline leg only
    S = min(args.cpu_sample, L)
    if membership:
        S = min(S, 3_000_000)
    num, den = synth.rows_per_position(num_docs)
    r0, r1 = synth.shard_rows(0, S, k, num, den, L)
    s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, num_docs)
    fn = oracle.membership if membership else oracle.conservation
    t0 = time.perf_counter()
    want = fn(s, e, o, 0, S, k, num_docs, literal=True)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(want, gpu_result_slice(S)))
    return {"value": S / dt, "unit": "query-positions/s", "cores": 1, "kind": "port",
            "sample": f"first {S} positions of the window ({r1 - r0} rows), literal bool-matrix "
                      f"port of memo_query.py:45-63,70 (oracle_literal_*), {dt:.1f} s",
            "parity_with_gpu_on_sample": ok}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    num_docs, L, membership = WORKLOADS[args.workload]
    k = args.k
    pivot = L * world
    qs, qe = rank * L, (rank + 1) * L
    ix, (r0, r1) = synth.device_index(qs, qe, k, num_docs, pivot, device=local)
    rows = r1 - r0
    W = (num_docs + 31) // 32
    if membership:
        out = torch.empty((L, W), dtype=torch.int32, device=dev)
        b_out = 4 * W
    else:
        out = torch.empty(L, dtype=torch.int16, device=dev)      # uint16 payload
        b_out = 2
    gathered = None
    if world > 1 and rank == 0:
        gathered = [torch.empty_like(out) for _ in range(world)]

    stream = torch.cuda.current_stream()

    def launch():
        if membership:
            ix.membership_dev(qs, qe, k, num_docs, out, stream.cuda_stream)
        else:
            ix.conservation_dev(qs, qe, k, num_docs, out, stream.cuda_stream)

    def step():
        launch()
        if world > 1:                      # result slices -> root over xGMI (RCCL send/recv)
            dist.gather(out, gathered, dst=0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ix.check(stream.cuda_stream)           # raises if the kernel flagged a bad row

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        launch()
        ev[i][1].record(stream)
        if world > 1:
            dist.gather(out, gathered, dst=0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        t = torch.tensor([dt, kern_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kern_ms = float(t[0]), float(t[1])
    ix.check(stream.cuda_stream)

    if rank == 0:
        b_alg = 24 * rows + b_out * L                  # 3 x int64 per row + output
        achieved = b_alg / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": "query-positions/sec (chr window, k=%d)" % k,
            "value": L * world * args.steps / dt,
            "unit": "query-positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"BASELINE config {args.workload[1:]}: synthetic {num_docs}-genome index, "
                                   f"{L} positions/GPU window, {rows} rows/GPU, k={k}, "
                                   f"{'membership' if membership else 'conservation'}",
                       "num_docs": num_docs, "window_per_gpu": L, "rows_per_gpu": rows, "k": k,
                       "query": "membership" if membership else "conservation",
                       "sharding": f"window split into {world} contiguous sub-windows, RCCL gather to rank 0"
                                   if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "sweep_membership_kernel" if membership else "sweep_conservation_kernel",
                         "kernel_ms": kern_ms, "algorithmic_bytes": b_alg},
        }
        prof = os.path.join(ROOT, "profiles", "traffic.json")    # PMC passes are separate runs
        if os.path.exists(prof):
            tj = json.load(open(prof)).get(args.workload)
            if tj:
                res["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
        if world == 1 and args.cpu_sample > 0:
            def gpu_slice(S):
                h = out[:S].cpu().numpy()
                return h.view(np.uint32) if membership else h.view(np.uint16)
            res["cpu_baseline"] = cpu_baseline(args, num_docs, L, k, membership, gpu_slice)
        print(json.dumps(res), flush=True)
    ix.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
