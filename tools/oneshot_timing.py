#!/usr/bin/env python3
"""H2D-inclusive timing of the one-shot host ABI (memo_conservation): upload + finalize + sweep +
download, host arrays in and out.  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memo_amd  # noqa: E402
from memo_amd import _lib, synth  # noqa: E402

os.environ["MEMO_TIMING"] = "1"          # the library prints the phases of every call on stderr
BIG_ONLY = "--big-only" in sys.argv      # BASELINE config 3 through the dense way in only (thread-count sweeps)
for n, L in ((100, 100_000_000),) if BIG_ONLY else ((10, 10_000_000), (100, 20_000_000), (100, 100_000_000)):
    num, den = synth.rows_per_position(n)
    r0, r1 = synth.shard_rows(0, L, 31, num, den, L)
    s, e, o = synth.host_rows(r0, r1 - r0, num, den, n)
    for rep in range(5 if BIG_ONLY else 3):
        t = time.perf_counter()
        out = memo_amd.conservation(s, e, o, 0, L, 31, n)
        dt = time.perf_counter() - t
    if BIG_ONLY:
        print(f"N={n} L={L}: dense way in {dt * 1e3:.1f} ms -> {L / dt:.3g} positions/s", flush=True)
        sys.exit(0)
    _lib.use_ab(True)                         # (the way-in switches live in the AB library: memo_debug_one_shot_way)
    _lib.check(_lib.lib().memo_debug_one_shot_way(2))   # the 4-byte words (round 2's way in) instead of the dense rows
    for rep in range(2):
        t = time.perf_counter()
        out_p = memo_amd.conservation(s, e, o, 0, L, 31, n)
        dt_p = time.perf_counter() - t
    _lib.check(_lib.lib().memo_debug_one_shot_way(0))
    assert np.array_equal(out, out_p)
    print(f"N={n} L={L}: 4-byte way in {dt_p * 1e3:.1f} ms", flush=True)
    _lib.check(_lib.lib().memo_debug_one_shot_way(1))   # the int64 way in, for comparison (round 1's only way)
    t = time.perf_counter()
    out_w = memo_amd.conservation(s, e, o, 0, L, 31, n)
    dt_w = time.perf_counter() - t
    _lib.check(_lib.lib().memo_debug_one_shot_way(0))
    _lib.use_ab(False)
    assert np.array_equal(out, out_w)
    gb = (s.nbytes * 3 + out.nbytes) / 1e9
    print(f"N={n} L={L}: {r1 - r0} rows ({gb:.2f} GB of int64 columns + result): one-shot {dt * 1e3:.1f} ms -> "
          f"{L / dt:.3g} positions/s (dense way in: 3.2 B per row on PCIe); int64 way in {dt_w * 1e3:.1f} ms -> {L / dt_w:.3g} positions/s",
          flush=True)

# ---- the pinned ring by itself: a packed index to host memory and back (memo_index_export_packed /
# memo_index_import_packed: pageable host arrays on both sides, worker threads + pinned slots + async copies)
import ctypes as C
from memo_amd import _lib
n, L = 100, 100_000_000
num, den = synth.rows_per_position(n)
r0, r1 = synth.shard_rows(0, L, 31, num, den, L)
ix, _ = synth.device_index(0, L, 31, n, L, pack="only")
inf = ix.info()
pk = np.empty(inf["rows"], np.uint32)
boff = np.empty(inf["buckets"], np.int64)
for rep in range(3):
    t = time.perf_counter()
    _lib.check(_lib.lib().memo_index_export_packed(ix._h, pk.ctypes.data, None, boff.ctypes.data, None))
    dt = time.perf_counter() - t
print(f"export (D2H through the pinned ring): {(pk.nbytes + boff.nbytes) / 1e9:.2f} GB in {dt * 1e3:.1f} ms = "
      f"{(pk.nbytes + boff.nbytes) / dt / 1e9:.1f} GB/s", flush=True)
for rep in range(3):
    h = C.c_void_p()
    t = time.perf_counter()
    _lib.check(_lib.lib().memo_index_import_packed(inf["rows"], 0, inf["bucket_shift"], 0, pk.ctypes.data, None, boff.ctypes.data,
                                                   len(boff), 0, inf["min_start"], inf["max_start"], inf["max_annot"], None, 0,
                                                   C.byref(h)))
    dt = time.perf_counter() - t
    _lib.lib().memo_index_destroy(h)
print(f"import (H2D through the pinned ring, incl. hipMalloc of the index): {(pk.nbytes + boff.nbytes) / 1e9:.2f} GB in "
      f"{dt * 1e3:.1f} ms = {(pk.nbytes + boff.nbytes) / dt / 1e9:.1f} GB/s", flush=True)
ix.close()
