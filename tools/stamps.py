#!/usr/bin/env python3
"""Per-phase cycle shares of the conservation sweep (diagnostic -DMEMO_STAMPS build only):
  MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_stamps.so python tools/stamps.py --workload c3 --pack only"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memo_amd import _lib, synth  # noqa: E402
from memo_amd.bench_legs import WORKLOADS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--pack", default=None)
ap.add_argument("--tuning", default="0,0,0")
ap.add_argument("--length", type=int, default=0, help="window length (default: the workload's); 10^7 positions of config 3 fit the Infinity Cache")
a = ap.parse_args()
num_docs, L, _ = WORKLOADS[a.workload]
L = a.length or L
_lib.use_ab(True)              # the stamps build is an AB build (make EXTRA=-DMEMO_STAMPS OUT_AB=...)
ix, _ = synth.device_index(0, L, a.k, num_docs, L, pack=a.pack)
out = torch.empty(L, dtype=torch.int16, device="cuda")
ix.debug_set_tuning(*[int(x) for x in a.tuning.split(",")])
names = ["locate tile", "issue loads + clear LDS + barrier", "wait for rows + scatter", "barrier after scatter", "fold",
         "store", "-", "tiles"]
nblocks = 8 * ((L // 128 + 8) // 8 + 1) + 8
buf = torch.zeros(8 * nblocks, dtype=torch.int64, device="cuda")
_lib.check(_lib.lib().memo_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())))
for rep in range(2):
    buf.zero_()
    ix.conservation_dev(0, L, a.k, num_docs, out)
    torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(-1, 8)
st = st[st[:, 7] == 1]
tiles = len(st)
tot = float(st[:, :6].sum())
print(f"{a.workload} k={a.k} pack={a.pack} tuning={a.tuning}: {tiles} tiles, {tot / max(tiles, 1):.0f} cycles per tile (wave 0)")
for i, n in enumerate(names[:6]):
    col = st[:, i].astype(np.float64)
    print(f"  {n:36s} {col.sum() / tot * 100:5.1f} %   mean {col.mean():8.0f}  median {np.median(col):8.0f} cycles/tile")
