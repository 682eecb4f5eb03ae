#!/usr/bin/env python3
"""The dense-row sweep variants -- table-driven (row_source 8, the product's), persistent with LDS-DMA / register staging
(4, 6, 7: AB library) -- against the round-2 kernel (row_source 5: every wave works its tile out): windows of many shapes on a
synthetic index, every k <= 64 that changes the number of level arrays.  A quick gate in front of an A/B run; the parity test
proper (with the oracle) is tests/test_gpu_parity.py::test_dense_row_sweep_variants.  GPU box only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memo_amd import _lib, synth  # noqa: E402

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
    print("p3 check:", "ok" if not bad else f"{bad} problems")
sys.exit(1 if bad else 0)
