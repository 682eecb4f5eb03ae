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

os.environ["MEMO_TIMING"] = "1"          # the library prints the phases of every call on stderr
for n, L in ((10, 10_000_000), (100, 20_000_000), (100, 100_000_000)):
    num, den = synth.rows_per_position(n)
    r0, r1 = synth.shard_rows(0, L, 31, num, den, L)
    s, e, o = synth.host_rows(r0, r1 - r0, num, den, n)
    for rep in range(3):
        t = time.perf_counter()
        out = memo_amd.conservation(s, e, o, 0, L, 31, n)
        dt = time.perf_counter() - t
    os.environ["MEMO_ONESHOT_WIDE"] = "1"     # the int64 way in, for comparison (round 1's only way)
    t = time.perf_counter()
    out_w = memo_amd.conservation(s, e, o, 0, L, 31, n)
    dt_w = time.perf_counter() - t
    del os.environ["MEMO_ONESHOT_WIDE"]
    assert np.array_equal(out, out_w)
    gb = (s.nbytes * 3 + out.nbytes) / 1e9
    print(f"N={n} L={L}: {r1 - r0} rows ({gb:.2f} GB of int64 columns + result): one-shot {dt * 1e3:.1f} ms -> "
          f"{L / dt:.3g} positions/s (packed way in); int64 way in {dt_w * 1e3:.1f} ms -> {L / dt_w:.3g} positions/s",
          flush=True)
