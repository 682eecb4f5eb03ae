#!/usr/bin/env python3
"""Timing of DAP -> BED rows on a synthetic DAP.  --reference runs /root/reference/src/dap_to_bed.py
(authoring container only: CPU, pure Python); default runs memo_amd.dap_to_bed on the GPU."""
import argparse
import io
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--positions", type=int, default=200_000)
ap.add_argument("--columns", type=int, default=99)
ap.add_argument("--out", default="/tmp/dap_timing")
ap.add_argument("--reference", action="store_true")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
dap, fai = os.path.join(a.out, "t.dap.txt"), os.path.join(a.out, "t.fa.fai")
rng = np.random.default_rng(1)
cur = rng.integers(1, 30, a.columns)
rows = np.empty((a.positions, a.columns + 1), np.int64)
rows[:, 0] = np.arange(a.positions)
for i in range(a.positions):                       # matching-statistic-like columns
    cur = np.maximum(cur - 1, 0)
    bump = rng.random(a.columns) < 0.1
    cur = np.where(bump, rng.integers(0, 60, a.columns), cur)
    rows[i, 1:] = np.minimum(cur, a.positions - i)
np.savetxt(dap, rows, fmt="%d", delimiter=" ")
open(fai, "w").write(f"chrT\t{a.positions}\t0\t60\t61\n")
print(f"DAP {a.positions} x {a.columns}: {os.path.getsize(dap) / 1e6:.0f} MB", flush=True)
if a.reference:
    t = time.time()
    r = subprocess.run([sys.executable, "/root/reference/src/dap_to_bed.py", "--mem", "--order", "--overlap", "--fai", fai,
                        "--dap", dap], capture_output=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    print(f"reference dap_to_bed.py: {time.time() - t:.1f} s, {len(r.stdout) / 1e6:.1f} MB of BED, rc {r.returncode}")
else:
    from memo_amd import dap_to_bed as D
    args = D.parse_arguments(["--mem", "--order", "--overlap", "--fai", fai, "--dap", dap])
    for rep in range(2):
        buf = io.BytesIO()
        t = time.time()
        D.main(args, buf)
        print(f"memo_amd.dap_to_bed (GPU): {time.time() - t:.2f} s, {len(buf.getvalue()) / 1e6:.1f} MB of BED")
    t = time.time()
    blocks = list(D.dap_blocks(dap))
    tp = time.time() - t
    names, rb = D.parse_fai(fai)
    t = time.time()
    with D.DapConverter(a.columns, rb, True, True) as conv:
        out = [conv.push(b[:, 1:]) for b in blocks]
    tg = time.time() - t
    print(f"  text parse {tp:.2f} s, GPU push+fetch {tg:.3f} s ({sum(len(o[0]) for o in out)} rows)")
