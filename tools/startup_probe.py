#!/usr/bin/env python3
"""Where a cache-hit `memo query` spends its first 0.1 s: library load, HIP runtime start, first allocation, an 80 MB upload from
pageable memory (runtime's copy vs the pinned ring).  ctypes only, one process per line.  GPU box only."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    mode = sys.argv[1]
    t0 = time.perf_counter()
    L = C.CDLL(os.path.join(ROOT, "memo_amd", "libmemo_amd.so"))
    t1 = time.perf_counter()
    L.memo_device_count()
    t2 = time.perf_counter()
    d = C.c_void_p()
    L.memo_dev_malloc.argtypes = [C.c_int32, C.c_size_t, C.POINTER(C.c_void_p)]
    L.memo_dev_malloc(0, 100 << 20, C.byref(d))
    t3 = time.perf_counter()
    buf = bytearray(80 << 20)
    src = (C.c_char * len(buf)).from_buffer(buf)
    t4 = time.perf_counter()
    if mode == "plain":
        L.memo_dev_upload.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.memo_dev_upload(0, d, src, len(buf), None)
    t5 = time.perf_counter()
    print(f"{mode}: dlopen {1e3 * (t1 - t0):.1f} ms, HIP start (memo_device_count) {1e3 * (t2 - t1):.1f} ms, first hipMalloc(100 MB) "
          f"{1e3 * (t3 - t2):.1f} ms, 80 MB pageable -> device {1e3 * (t5 - t4):.1f} ms")
else:
    for mode in ("plain", "plain", "none"):
        t = time.perf_counter()
        r = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True)
        print(r.stdout.strip(), f"| process wall {1e3 * (time.perf_counter() - t):.0f} ms", r.stderr.strip()[-200:] if r.returncode else "")
    t = time.perf_counter()
    subprocess.run([sys.executable, "-c", "pass"])
    print(f"python -c pass: {1e3 * (time.perf_counter() - t):.0f} ms")
