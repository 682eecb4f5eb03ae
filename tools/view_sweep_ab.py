#!/usr/bin/env python3
"""Round 5: the conservation sweep on a dense k-class view as memo_view.hip builds it: five and six rows per group, rows placed and
not.  (With round 4's builder still in the tree -- places chosen after a sort by the first block's bank -- its views measured 0.6 %
faster at k = 31 than the fused pass's placed ones, level at k = 21: profiles/r05_view_pass.txt.)  One index per variant,
`--launches` launches back to back, median of the last two thirds; variants alternate `--reps` times.  GPU box; A/B library."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="31")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--launches", type=int, default=900)
    ap.add_argument("--variants", default="0:5:1,0:5:0,0:6:1,0:6:0", help="0:rows_per_group:placed, ...")
    a = ap.parse_args()
    import numpy as np
    import torch
    from memo_amd import _lib, synth
    _lib.use_ab(True)
    lib = _lib.lib()
    num_docs, L = 100, 100_000_000
    out = torch.empty(L, dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream()
    ix, (r0, r1) = synth.device_index(0, L, 31, num_docs, L, pack="dense")
    with ix:
        for rep in range(a.reps):
            for var in a.variants.split(","):
                builder, rpg, placed = (int(x) for x in var.split(":"))
                lib.memo_debug_view_colouring(placed)
                lib.memo_debug_six_views(1 if rpg == 6 else 0)
                for k in [int(x) for x in a.ks.split(",")]:
                    ix.set_option(1, 0)
                    ix.set_option(1, 1)
                    ix.prepare(k, num_docs)
                    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.launches)]
                    for e0, e1 in ev:
                        e0.record(stream)
                        ix.conservation_u8_dev(0, L, k, num_docs, out, stream.cuda_stream)
                        e1.record(stream)
                    torch.cuda.synchronize()
                    ms = np.array([e0.elapsed_time(e1) for e0, e1 in ev[a.launches // 3:]])
                    inf = ix.info()
                    print(json.dumps({"rows_per_group": rpg, "placed": bool(placed), "k": k,
                                      "ms_median": round(float(np.median(ms)), 4), "ms_min": round(float(ms.min()), 4),
                                      "rows_read": inf["last_rows_read"], "variant": inf["last_variant"], "view_ms": round(inf["last_view_ms"], 3)}),
                          flush=True)
        lib.memo_debug_view_colouring(1)
        lib.memo_debug_six_views(-1)


if __name__ == "__main__":
    main()
