#!/usr/bin/env python3
"""Modelled weak scaling of bench.py --gpus N on one node (NOT a measurement: this pool gives one GPU).

Step time = the slowest of: a peer (sweep + encode of its slice), rank 0 (its own sweep of `root_weight`
of a share + decoding the world - 1 slices it received; its own slice never travels), a peer's xGMI link to
rank 0 (wire bytes / link rate; gather i overlaps sweep i + 1) -- memo_amd.shard.modelled_step, with the
per-kernel figures measured on one MI355X (defaults: profiles/r02_*; override on the command line).

    python tools/scaling_model.py [--sweep-ms 0.37] [--link-GBs 75] ...
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memo_amd import shard  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=100_000_000, help="positions per GPU (config 3)")
ap.add_argument("--sweep-ms", type=float, default=0.332)
ap.add_argument("--link-GBs", type=float, default=75.0, help="one direction of one xGMI link")
ap.add_argument("--nibble", default="53101352,0.031,0.108", help="wire bytes, decode ms per slice, encode ms")
ap.add_argument("--dense", default="41250080,0.058,0.153")
ap.add_argument("--runs", default="23046320,0.036,0.064")
ap.add_argument("--runs-many", default="", help="ms to decode 1, 3 and 7 slices of the runs coding with ONE launch (memo_transport_runs_unpack_many_dev), "
                                                "comma-separated; empty: world - 1 launches of the single-slice decode")
ap.add_argument("--value-bytes", type=int, default=1, help="bytes per result value as plain bytes (2: more than 255 genomes, config 5)")
ap.add_argument("--only", default="", help="comma-separated codings to consider besides plain (config 5: runs)")
a = ap.parse_args()
codings = {"plain": (a.L * a.value_bytes, 0.0, 0.0)}
for name in ("nibble", "dense", "runs"):
    if a.only and name not in a.only.split(","):
        continue
    b, d, e = getattr(a, name).split(",")
    codings[name] = (int(b), float(d) * 1e-3, float(e) * 1e-3)
sweep, link = a.sweep_ms * 1e-3, a.link_GBs * 1e9
single = a.L / sweep
print(f"one GPU: {a.sweep_ms} ms per {a.L} positions = {single:.3g} positions/s; link {a.link_GBs} GB/s per peer (assumed)")
print("N  root_weight  coding   step_ms   positions/s   x one GPU   bound by")
for world in (2, 4, 8):
    best = None
    for w in shard.ROOT_WEIGHTS + (0.0,):
        for name, (wire, dec, enc) in codings.items():
            peer = sweep + enc
            many = [float(x) * 1e-3 for x in a.runs_many.split(",")] if (a.runs_many and name == "runs") else None
            root = w * sweep + (many[{2: 0, 4: 1, 8: 2}[world]] if many else (world - 1) * dec)
            wire_t = wire / link
            step = max(peer, root, wire_t)
            total = (world - 1 + w) * a.L / step
            row = (total, world, w, name, step, "peer sweep+encode" if step == peer else ("root sweep+decode" if step == root else "link"))
            if w == 1.0 and name == "plain":
                print(f"{world}  {w:<11} {name:<8} {step * 1e3:7.3f}   {total:11.3g}   {total / single:6.2f}     {row[5]}")
            if best is None or total > best[0]:
                best = row
    total, world, w, name, step, why = best
    print(f"{world}  {w:<11} {name:<8} {step * 1e3:7.3f}   {total:11.3g}   {total / single:6.2f}     {why}   <- best of the grid")
