#!/usr/bin/env python3
"""The bucket-grouped form of the k-class views (row_source 13: six 18-bit rows per 16 bytes, sweep_conservation_six_kernel) against
the sweep of all the dense rows (row_source 9): every view class, windows of many shapes, both result types.  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memo_amd import _lib, synth  # noqa: E402

_lib.use_ab(True)
n, L = 100, 12_000_000
ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack="dense")
bad = 0
with ix:
    for k in list(range(2, 35)):
        for i, (qs, qe) in enumerate(((0, L), (4, L - 3), (1_000_000, 9_000_001), (2_345_676, 5_000_000), (0, L), (8, 100_000), (11_999_000, L + 500))):
            for dt in (np.uint8, np.uint16):
                ix.debug_set_tuning(0, 0, 0, 9, 0)
                ref = ix.conservation(qs, qe, k, n, dtype=dt)
                ix.debug_set_tuning(0, 0, 0, 13, 0)
                got = ix.conservation(qs, qe, k, n, dtype=dt)
                inf = ix.info()
                if not np.array_equal(ref, got):
                    bad += 1
                    d = np.flatnonzero(ref != got)
                    print("MISMATCH", k, qs, qe, dt.__name__, inf["last_sweep"], inf["last_variant"], len(d), d[:8], ref[d[:8]], got[d[:8]])
        print("k", k, "variant", ix.info()["last_variant"], "rows read", ix.info()["last_rows_read"], flush=True)
    print("six check:", "ok" if not bad else f"{bad} problems")
sys.exit(1 if bad else 0)
