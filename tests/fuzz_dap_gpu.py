#!/usr/bin/env python3
"""Time-boxed differential fuzzing of the GPU index-row construction (memo_dap_*) against the
oracle restatement of dap_to_bed.py (GPU box):  python tests/fuzz_dap_gpu.py --seconds 120"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memo_amd.dap_to_bed import DapConverter  # noqa: E402
from oracle import dap_oracle as O  # noqa: E402  (the checker)

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60)
ap.add_argument("--seed", type=int, default=int(time.time()))
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
print("seed", a.seed, flush=True)
t_end, cases = time.time() + a.seconds, 0
while time.time() < t_end:
    cases += 1
    C_ = int(rng.choice([1, 2, 3, 31, 32, 33, 64, 65, 99, 128, 129, 499, 700]))
    nrec = int(rng.integers(1, 6))
    lens = rng.integers(1, int(rng.choice([3, 50, 2000])), nrec)
    rec_begin = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(rec_begin[-1])
    npos = total if rng.random() < 0.7 else int(rng.integers(1, total + 1))      # sometimes truncated
    hi = int(rng.choice([2, 10, 60, 5000]))
    lcp = rng.integers(0, hi, (npos, C_)).astype(np.int32)
    if rng.random() < 0.5:                                   # matching-statistic-like: decays by one
        for i in range(1, npos):
            keep = rng.random(C_) < 0.8
            lcp[i] = np.where(keep, np.maximum(lcp[i - 1] - 1, 0), lcp[i])
    order, overlap = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    want = O.dap_rows(lcp, rec_begin, overlap, order)
    pieces = int(rng.choice([1, 2, 5, 17]))
    with DapConverter(C_, rec_begin, order, overlap) as conv:
        got = [conv.push(p) for p in np.array_split(lcp, pieces)] + [conv.finish()]
    cat = [np.concatenate([g[i] for g in got]) for i in range(4)]
    if not all(np.array_equal(x, y) for x, y in zip(cat, want)):
        print("MISMATCH", dict(seed=a.seed, case=cases, C=C_, lens=lens.tolist(), npos=npos, order=order, overlap=overlap, pieces=pieces))
        np.savez("/tmp/fuzz_dap_fail.npz", lcp=lcp, rec_begin=rec_begin)
        sys.exit(1)
print(f"dap fuzz ok: {cases} DAPs in {a.seconds:.0f} s", flush=True)
