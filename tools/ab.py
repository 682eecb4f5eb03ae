#!/usr/bin/env python3
"""Interleaved A/B timing of kernel tunings in ONE process on ONE device (box-to-box spread
on this pool is larger than most tuning effects).  GPU box only.

  python tools/ab.py --workload c3 --k 31 --rounds 12 "0,0,0" "1024,1,0" "4096,4,0"
each variant = tile_w,waves,membership_algo[,row_source[,scatter]]  (memo_debug_set_tuning, per index;
libmemo_amd_ab.so)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memo_amd import _lib, synth  # noqa: E402
from memo_amd.bench_legs import WORKLOADS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--u8", action="store_true")
    ap.add_argument("--length", type=int, default=0, help="query [0, length) instead of the whole pivot")
    ap.add_argument("--qs", type=int, default=0, help="the window starts here (results of [qs, L))")
    ap.add_argument("--num-docs", type=int, default=0, help="this many genomes instead of the workload's (same rows per genome and position)")
    ap.add_argument("--density", default="5/100", help="rows per genome and position")
    ap.add_argument("--pack", default=None, choices=[None, "keep", "only", "dense", "both"],
                    help="keep: int64 + 4-byte rows; only: 4-byte rows; dense: dense rows only; both: 4- and dense rows "
                         "(the library reads the dense ones where they can answer; row_source 3 in a variant selects the 4-byte rows)")
    ap.add_argument("--row-order", type=int, default=0,
                    help="order of the 4-byte rows inside a bucket (memo_debug_row_order): 0 library's, 1 start order, 2 chunks dealt "
                         "over the starts, 3 + by overlap mod 32")
    ap.add_argument("--rows-file", default=None,
                    help="an index from a file (.npz with start, end, annot, num_docs, length: tools/realistic_index.py) instead of the "
                         "synthetic generator; the window is [0, length)")
    ap.add_argument("--membership", action="store_true", help="with --rows-file: membership queries")
    ap.add_argument("--six", action="store_true", help="dense k-class views as groups of six rows (memo_debug_six_views 1)")
    ap.add_argument("--no-colour", action="store_true", help="dense k-class views keep the order the filter leaves (memo_debug_view_colouring 0)")
    ap.add_argument("--prepare", action="store_true", help="memo_index_prepare first: the k-class view and the order of the rows are there "
                                                           "before the first timed launch (otherwise the library's rule may build them mid-run)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    num_docs, L, membership = WORKLOADS[a.workload]
    num_docs = a.num_docs or num_docs
    L = a.length or L
    from fractions import Fraction
    _lib.use_ab(True)
    if a.no_colour:
        _lib.lib().memo_debug_view_colouring(0)
    if a.six:
        _lib.lib().memo_debug_six_views(1)
    if a.rows_file:
        import memo_amd
        z = np.load(a.rows_file)
        num_docs, L, membership = int(z["num_docs"]), int(z["length"]), a.membership
        ix = memo_amd.DeviceIndex.from_host(*(np.ascontiguousarray(z[c], dtype=np.int64) for c in ("start", "end", "annot")))
        r0, r1 = 0, len(z["start"])
        if a.pack:
            ix.pack(keep_wide=(a.pack == "keep"))
        if a.pack == "dense":
            ix.pack_dense(keep_packed=False)
    else:
        ix, (r0, r1) = synth.device_index(0, L, a.k, num_docs, L, density=Fraction(a.density),
                                          pack="only" if a.pack == "both" else a.pack)
    if a.row_order:
        ix.debug_row_order(a.row_order)
    if a.pack == "both":
        ix.pack_dense(keep_packed=True)
    if a.prepare:
        ix.prepare(a.k, num_docs, membership=membership)
    W = (num_docs + 31) // 32
    out = torch.empty((L, W) if membership else (L,), dtype=torch.int32 if membership else torch.int16, device="cuda")
    st = torch.cuda.current_stream()
    variants = [tuple(int(x) for x in v.split(",")) for v in a.variants]
    times = {v: [] for v in variants}

    def launch():
        if membership:
            ix.membership_dev(a.qs, L, a.k, num_docs, out, st.cuda_stream)
        elif a.u8:
            ix.conservation_u8_dev(a.qs, L, a.k, num_docs, out, st.cuda_stream)
        else:
            ix.conservation_dev(a.qs, L, a.k, num_docs, out, st.cuda_stream)

    # Every variant's RESULT against the first variant's before a time is printed (round 4: the first ds_write_addtid_b32 clear
    # was wrong AND faster, and this harness, which timed without checking, said "faster").  A checksum per launch:
    # the sum of the result's 64-bit words, on the device, one number back.
    def checksum():
        flat = out.reshape(-1).view(torch.uint8)
        n8 = flat.numel() // 8 * 8
        return int(flat[:n8].view(torch.int64).sum().item()) ^ (int(flat[n8:].to(torch.int64).sum().item()) << 1)

    sums = {}
    for r in range(a.rounds + 1):
        for v in variants:
            ix.debug_set_tuning(*(list(v) + [0] * 5)[:5])
            if r < 2:
                out.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            launch()
            e1.record(st)
            torch.cuda.synchronize()
            if r < 2:                               # (the warm-up round and the first timed one)
                sums.setdefault(v, set()).add(checksum())
            if r:                                   # round 0 = warm-up
                times[v].append(e0.elapsed_time(e1))
    ix.check()
    want = sums[variants[0]]
    for v in variants:
        if sums[v] != want or len(sums[v]) != 1:
            raise SystemExit(f"ab.py: variant {v} gives another result than variant {variants[0]} (checksums {sorted(sums[v])} against "
                             f"{sorted(want)}): no time is printed for a kernel that is wrong")
    for v in variants:
        src = v[3] if len(v) > 3 else 0
        brow = 24 if (not a.pack or src == 1) else (3.2 if ix.info()["dense_rows"] and (src == 2 or a.pack == "dense") and a.k <= 64 and not membership
                                                    else (6 if ix.info()['packed_format'] == 6 else 4))
        b_alg = brow * (r1 - r0) + (4 * W if membership else (1 if a.u8 else 2)) * L
        t = np.array(times[v])
        inf = ix.info()
        print(json.dumps({"variant": v, "workload": a.rows_file or a.workload, "k": a.k, "row_bytes": brow, "ms_median": float(np.median(t)),
                          "last_sweep": inf["last_sweep"], "last_rows_read": inf["last_rows_read"], "level_arrays": inf["last_level_arrays"],
                          "ms_min": float(t.min()), "ms_max": float(t.max()),
                          "frac_of_8TBs": b_alg / (float(np.median(t)) * 1e-3) / 8e12}))


if __name__ == "__main__":
    main()
