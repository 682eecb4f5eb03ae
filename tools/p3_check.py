#!/usr/bin/env python3
"""The persistent dense-row sweep (row_source 4) against the tile-per-workgroup kernel (row_source 5) and the oracle:
windows of many shapes on a synthetic index, every k <= 64 that changes the number of level arrays.  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memo_amd import _lib, synth  # noqa: E402
from oracle import memo_oracle as oracle  # noqa: E402  (checker)

_lib.use_ab(True)
n, L = 100, 30_000_000
ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack="dense")
num, den = synth.rows_per_position(n)
bad = 0
with ix:
    for k in (2, 3, 4, 5, 8, 9, 16, 17, 21, 31, 32, 33, 48, 64):
        for qs, qe in ((0, L), (4, L - 3), (1_000_000, 29_000_001), (12_345_676, 25_000_000)):
            for dt in (np.uint8, np.uint16):
                ix.debug_set_tuning(0, 0, 0, 5, 0)
                ref = ix.conservation(qs, qe, k, n, dtype=dt)
                v_ref = ix.info()["last_variant"]
                for src in (8, 4, 6, 7):
                    ix.debug_set_tuning(0, 0, 0, src, 0)
                    got = ix.conservation(qs, qe, k, n, dtype=dt)
                    inf = ix.info()
                    same = np.array_equal(ref, got)
                    if not same or inf["last_variant"] != (2 if src == 8 else 1) or v_ref != 0:
                        bad += 1
                        d = np.flatnonzero(ref != got)
                        print("MISMATCH" if not same else "variant?", src, k, qs, qe, dt.__name__, inf["last_sweep"], inf["last_variant"],
                              v_ref, len(d), d[:8], ref[d[:8]], got[d[:8]])
        a, b = 7_000_000, 7_400_000
        sr0, sr1 = synth.shard_rows(a, b, k, num, den, L)
        s, e, o = oracle.synth_rows(sr0, sr1 - sr0, num, den, n)
        want = oracle.conservation(s, e, o, a, b, k, n, literal=False)
        ix.debug_set_tuning(0, 0, 0, 8, 0)
        full = ix.conservation(0, L, k, n)
        if not np.array_equal(full[a:b], want):
            bad += 1
            print("ORACLE MISMATCH", k)
    print("p3 check:", "ok" if not bad else f"{bad} problems")
sys.exit(1 if bad else 0)
