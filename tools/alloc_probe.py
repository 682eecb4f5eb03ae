#!/usr/bin/env python3
"""Round 4: is the k >= 65 sweep's two-state timing (profiles/r04_large_k.txt) a property of the PROCESS or of where the index
landed in HBM?  One process builds the same index several times (a dummy allocation of another size in front of each), and
times the same query on each.  GPU box only.

  python tools/alloc_probe.py --workload c5 --k 101 --trials 6
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memo_amd import _lib, synth  # noqa: E402
from memo_amd.bench_legs import WORKLOADS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c5")
    ap.add_argument("--k", type=int, default=101)
    ap.add_argument("--trials", type=int, default=6)
    ap.add_argument("--launches", type=int, default=3000)
    ap.add_argument("--u8", action="store_true")
    ap.add_argument("--pack", default="only", choices=["only", "dense"])
    ap.add_argument("--same-place", action="store_true", help="no dummy allocations: every trial reuses the freed blocks")
    a = ap.parse_args()
    num_docs, L, membership = WORKLOADS[a.workload]
    _lib.use_ab(True)
    st = torch.cuda.current_stream()
    out = torch.empty((L,), dtype=torch.int16, device="cuda")
    pads = []
    for trial in range(a.trials):
        if not a.same_place:
            pads.append(torch.empty(((trial * 37 + 5) << 20) + trial * 4096, dtype=torch.uint8, device="cuda"))
        ix, (r0, r1) = synth.device_index(0, L, a.k, num_docs, L, pack=a.pack)
        ix.prepare(a.k, num_docs, False, L)
        fn = ix.conservation_u8_dev if a.u8 else ix.conservation_dev
        blocks = []
        for b in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(50):
                fn(0, L, a.k, num_docs, out, st.cuda_stream)
            e0.record(st)
            for _ in range(a.launches // 4):
                fn(0, L, a.k, num_docs, out, st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            blocks.append(e0.elapsed_time(e1) / (a.launches // 4))
        ix.check()
        print(json.dumps({"trial": trial, "workload": a.workload, "k": a.k, "rows": r1 - r0, "ms_blocks": [round(x, 4) for x in blocks],
                          "pad_bytes": 0 if a.same_place else pads[-1].numel(), "level_arrays": ix.info()["last_level_arrays"]}), flush=True)
        del ix
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
