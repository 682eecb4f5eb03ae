#!/usr/bin/env python3
"""The reference's seam (memo_conservation: host int64 columns in, host uint16 result out) on BASELINE config 3, for one
host thread count (MEMO_HOST_THREADS, read once per process): the columns first-touched by the library's pool threads
(what bench.py's download gives) and by ONE thread (what a NumPy caller has), the calls' phases on stderr (MEMO_TIMING),
the result compared with the resident path's; then the same rows as the reference's own [M, 3] array through
memo_conservation_rows.  GPU box.  usage: oneshot_sweep.py [calls]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memo_amd  # noqa: E402
from memo_amd import _lib, synth  # noqa: E402

os.environ["MEMO_TIMING"] = "1"
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n, L, k = 100, 100_000_000, 31
lib = _lib.lib()
allowed, quota = C.c_int32(), C.c_double()
threads = lib.memo_host_threads(C.byref(allowed), C.byref(quota))
print(f"host threads {threads} (CPUs allowed {allowed.value}, cgroup quota {quota.value:.1f} CPUs, MEMO_HOST_THREADS="
      f"{os.environ.get('MEMO_HOST_THREADS', '-')})", flush=True)
ix, (r0, r1) = synth.device_index(0, L, k, n, L)
rows = r1 - r0
want = ix.conservation(0, L, k, n)
ds, de, do = ix.columns()
pool_cols = [np.empty(rows, np.int64) for _ in range(3)]
for h, d in zip(pool_cols, (ds, de, do)):
    _lib.check(lib.memo_dev_download(0, h.ctypes.data, d, h.nbytes, None))
ix.close()
for label, cols in (("pool-touched", pool_cols), ("one-thread-touched", None)):
    if cols is None:
        cols = [c.copy() for c in pool_cols]      # NumPy's copy: every page first written by this thread
        del pool_cols
    ms = []
    for _ in range(calls):
        t = time.perf_counter()
        out = memo_amd.conservation(cols[0], cols[1], cols[2], 0, L, k, n)
        ms.append((time.perf_counter() - t) * 1e3)
    ok = bool(np.array_equal(out, want))
    print(f"{label}: calls {' '.join(f'{x:.1f}' for x in ms)} ms -> best {L / min(ms) * 1e3:.3g} positions/s, parity {ok}", flush=True)
    assert ok
# the reference's own array: [M, 3] row-major (memo_conservation_rows) -- made here from the columns (one thread: NumPy)
t = time.perf_counter()
aos = np.empty((rows, 3), np.int64)
for j in range(3):
    aos[:, j] = cols[j]
del cols
print(f"[M, 3] array made in {time.perf_counter() - t:.1f} s (NumPy, strided; what the reference's concat of pandas frames gives for free)", flush=True)
ms = []
for _ in range(calls):
    t = time.perf_counter()
    out = memo_amd.conservation_rows(aos, 0, L, k, n)
    ms.append((time.perf_counter() - t) * 1e3)
ok = bool(np.array_equal(out, want))
print(f"rows form: calls {' '.join(f'{x:.1f}' for x in ms)} ms -> best {L / min(ms) * 1e3:.3g} positions/s, parity {ok}", flush=True)
assert ok
