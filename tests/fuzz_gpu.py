#!/usr/bin/env python3
"""Time-boxed differential fuzzing of the HIP path against the oracle (GPU box):
    python tests/fuzz_gpu.py --seconds 300 [--seed S]
Random indexes (density, clumping, annots, overlaps incl. end < start), random windows, k, N,
tile shapes, membership algorithms, row formats, scatters (incl. the level plan of the mixed arrays), row orders, memo_index_prepare.  Exits non-zero on the first
mismatch and prints the case."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import memo_amd  # noqa: E402
from memo_amd import _lib  # noqa: E402
from oracle import memo_oracle as oracle  # noqa: E402  (the checker)

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60)
ap.add_argument("--seed", type=int, default=int(time.time()))
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
L = _lib.use_ab(True)          # libmemo_amd_ab.so: the product objects + per-index kernel-shape switches
t_end = time.time() + a.seconds
cases = queries = 0
print("seed", a.seed, flush=True)
while time.time() < t_end:
    cases += 1
    n_docs = int(rng.choice([2, 5, 31, 32, 33, 64, 100, 255, 256, 257, 500, 511, 512, 1000]))
    length = int(rng.choice([100, 3000, 50_000, 400_000]))
    m = int(rng.integers(0, int(rng.choice([50, 5000, 200_000]))))
    mode = int(rng.integers(0, 5))
    if mode == 0:
        s = rng.integers(1, length, m)
    elif mode == 1:
        s = rng.choice(rng.integers(1, length, max(int(rng.integers(1, 40)), 1)), m)
    elif mode == 2:
        s = rng.integers(-100, length + 1000, m)
    elif mode == 3:
        s = (rng.integers(1, max(length // 64, 2), m) * 64 + rng.integers(-1, 2, m))
    else:
        s = np.abs(rng.normal(length / 2, length / 20 + 1, m)).astype(np.int64) + 1
    s = np.sort(s).astype(np.int64)
    e = s + rng.integers(0, int(rng.choice([2, 30, 70, 300, 3000])), m)
    if rng.random() < 0.25 and m:
        neg = rng.random(m) < 0.02
        e[neg] = s[neg] - rng.integers(1, 2000, int(neg.sum()))
    hi_annot = n_docs if rng.random() < 0.9 else n_docs + 3          # sometimes outside the matrix
    o = rng.integers(0 if rng.random() < 0.1 else 1, max(hi_annot, 2), m).astype(np.int64)
    bshift = int(rng.choice([0, 0, 0, 1, 3, 6, 8]))
    packable = not (m and s.min() < 0)
    # a third of the packable indexes come in the packed, pinned way (memo_builder_*: host packer + bucket table
    # built on the host), in ragged pieces; the rest as int64 columns
    via_builder = packable and m > 0 and int(o.max()) <= 4095 and int(o.min()) >= 0 and rng.random() < 0.4
    # ... a third of those as DENSE rows straight from the host packer (five rows per 16 bytes; such an index answers k <= 64
    # only, may leave out the rows that can never write, and builds k-class views and tile tables behind its queries)
    dense_only = via_builder and int(o.max()) <= 511 and n_docs <= 511 and rng.random() < 0.4      # (annots of up to nine bits)
    if via_builder:
        cuts = [0] + sorted(int(x) for x in rng.integers(0, m, int(rng.integers(0, 4)))) + [m]
        with memo_amd.IndexBuilder(m + int(rng.integers(0, 100)), bucket_shift=bshift, dense=dense_only) as b:
            for a_, z_ in zip(cuts[:-1], cuts[1:]):
                b.push(s[a_:z_], e[a_:z_], o[a_:z_])
            made = b.finish()
    else:
        made = memo_amd.DeviceIndex.from_host(s, e, o, bucket_shift=bshift)
    with made as ix:
        def dense_fits():                                          # (format 12 with annots of nine bits: 256 .. 511 genomes)
            inf = ix.info()
            return inf["packed_format"] == 4 or (inf["packed_format"] == 12 and inf["max_annot"] <= 511)
        if via_builder and not dense_only:
            if dense_fits() and rng.random() < 0.5:
                ix.pack_dense(keep_packed=True)
        elif packable and rng.random() < 0.7:
            ix.pack(keep_wide=True)
            if dense_fits() and rng.random() < 0.6:
                ix.pack_dense(keep_packed=True)
        if rng.random() < 0.5:                                     # the order of the 4-byte rows inside their buckets (memo_interleave.hip):
            ix.debug_row_order(int(rng.integers(1, 5)))            #   start order, the two dealt orders, the membership order
        _lib.lib().memo_debug_view_colouring(int(rng.choice([0, 1, 1])))   # the places of a dense view's rows inside their groups
        _lib.lib().memo_debug_six_views(int(rng.choice([-1, -1, 0, 1])))             # which kind of dense view: the library's choice, five rows per group, six (memo_view.hip)
        k_pet = int(rng.choice([5, 9, 17, 31, 33, 101]))          # (asked often enough for its class's view to be built)
        if rng.random() < 0.3:                                     # memo_index_prepare: the view / tile table / row order before the first query
            try:
                ix.prepare(k_pet, n_docs, membership=bool(rng.random() < 0.3), window_hint=int(rng.choice([0, 1000, length])))
            except memo_amd.MemoError as exc:                      # (what the query itself would say: the dense rows alone cannot
                if not (dense_only and ("needs the" in str(exc) or "dropped" in str(exc))):        # answer this k or this index)
                    raise
        for _ in range(12):
            queries += 1
            k = k_pet if rng.random() < 0.45 else int(rng.choice(
                [1, 2, 3, 5, 8, 9, 16, 17, 21, 31, 32, 33, 64] + ([] if dense_only else [65, 101, 128, 129, 255, 256]) +
                ([] if via_builder else [257, 1000])))                       # (a packed-only index answers k <= 256, dense rows k <= 64)
            qs = int(rng.integers(0, length))
            if rng.random() < 0.5:
                qs &= ~3                                                      # (windows on the 4-position raster: aligned result stores)
            qe = int(rng.integers(qs, length + 200))
            tune = (int(rng.choice([0, 256, 512, 1024, 2048, 4096, 1472, 1728])), int(rng.choice([0, 1, 4, 8])),
                    int(rng.choice([0, 2, 3, 4])), int(rng.choice([0, 0, 0, 1, 2, 3, 5, 8, 9, 10, 13])), int(rng.integers(0, 6)))
            if dense_only and tune[3] in (1, 3):
                tune = tune[:3] + (0,) + tune[4:]                             # (no int64 columns / 4-byte rows to force)
            ix.debug_set_tuning(*tune)
            memb = rng.random() < 0.4
            if memb and (qe - qs) * n_docs > 30_000_000:
                qe = qs + 30_000_000 // n_docs
            rows = oracle.filter_rows(s, e, o, qs, qe, k)
            fn_o = oracle.membership if memb else oracle.conservation
            fn_g = ix.membership if memb else ix.conservation
            try:
                want, werr = fn_o(*rows, qs, qe, k, n_docs, literal=False), None
            except IndexError:
                want, werr = None, IndexError
            try:
                got, gerr = fn_g(qs, qe, k, n_docs), None
            except IndexError:
                got, gerr = None, IndexError
            except memo_amd.MemoError as exc:
                if dense_only and "needs the" in str(exc):     # (sparse index, tile shape, ...: the dense rows alone cannot answer;
                    continue                                    #  the product's callers ask memo_dense_rows_can_answer first)
                raise
            ok = werr == gerr and (werr is not None or np.array_equal(got, want))
            if not ok:
                print("MISMATCH", dict(seed=a.seed, case=cases, via_builder=via_builder, n_docs=n_docs, length=length, m=m, mode=mode, k=k, qs=qs, qe=qe,
                                       tune=tune, memb=memb, info=ix.info(), werr=str(werr), gerr=str(gerr)), flush=True)
                np.savez("/tmp/fuzz_fail.npz", s=s, e=e, o=o)
                sys.exit(1)
print(f"fuzz ok: {cases} indexes, {queries} queries in {a.seconds:.0f} s", flush=True)
