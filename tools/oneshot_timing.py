#!/usr/bin/env python3
"""H2D-inclusive timing of the one-shot host ABI (memo_conservation): upload + finalize + sweep +
download, host arrays in and out.  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memo_amd  # noqa: E402
from memo_amd import synth  # noqa: E402

for n, L in ((10, 10_000_000), (100, 20_000_000), (100, 100_000_000)):
    num, den = synth.rows_per_position(n)
    r0, r1 = synth.shard_rows(0, L, 31, num, den, L)
    s, e, o = synth.host_rows(r0, r1 - r0, num, den, n)
    for rep in range(2):
        t = time.perf_counter()
        out = memo_amd.conservation(s, e, o, 0, L, 31, n)
        dt = time.perf_counter() - t
    gb = (s.nbytes * 3 + out.nbytes) / 1e9
    print(f"N={n} L={L}: {r1 - r0} rows, {gb:.2f} GB over PCIe, one-shot {dt * 1e3:.1f} ms -> {L / dt:.3g} positions/s "
          f"({gb / dt:.1f} GB/s host<->device incl. finalize + sweep)", flush=True)
