#!/usr/bin/env python3
"""Round 5: what building a dense k-class view costs (memo_index_prepare -> info.last_view_ms: HIP events around the whole pass on its
stream, allocations and the two host waits included), round 4's five kernels against the fused pass of memo_view.hip, views of five
and of six rows per group.  BASELINE config 3 (or --workload c5: 500 genomes, nine-bit annots).  GPU box; A/B library."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"])
    ap.add_argument("--ks", default="17,21,31")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--builders", default="0", help="(round 4's builder, 1, is gone: its timings are in profiles/r05_view_pass.txt)")
    ap.add_argument("--colour", type=int, default=1, help="0: the rows of a view keep the order they come in (memo_debug_view_colouring)")
    a = ap.parse_args()
    import torch  # noqa: F401  (load order: INTEGRATION.md section 5)
    from memo_amd import _lib, synth
    _lib.use_ab(True)
    lib = _lib.lib()
    lib.memo_debug_view_colouring(a.colour)
    num_docs, L = (100, 100_000_000) if a.workload == "c3" else (500, 1 << 25)
    ix, (r0, r1) = synth.device_index(0, L, 31, num_docs, L, pack="dense")
    with ix:
        for rep in range(a.reps):
            for builder in [int(x) for x in a.builders.split(",")]:
                for six in ((0, 1) if num_docs <= 255 else (0,)):
                    lib.memo_debug_six_views(six)
                    for k in [int(x) for x in a.ks.split(",")]:
                        ix.set_option(1, 0)          # MEMO_OPT_VIEWS off and on again: the views go
                        ix.set_option(1, 1)
                        taken = ix.prepare(k, num_docs)
                        inf = ix.info()
                        print(json.dumps({"workload": a.workload, "builder": "round 4" if builder else "fused", "placed": bool(a.colour), "rows_per_group": 6 if six else 5,
                                          "k": k, "view_ms": round(inf["last_view_ms"], 3), "rows_in": r1 - r0, "bytes_taken": taken,
                                          "views": inf["views_resident"]}), flush=True)
        lib.memo_debug_six_views(0)


if __name__ == "__main__":
    main()
