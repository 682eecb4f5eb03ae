#!/usr/bin/env python3
"""Round 5: what the gather's coding costs per step on one MI355X -- the sweep, the runs coding's encoder (a peer), and the decode of
the 1 / 3 / 7 slices rank 0 receives at 2 / 4 / 8 GPUs, one launch per slice against ONE launch for all
(memo_transport_runs_unpack_many_dev) -- for tools/scaling_model.py.  HIP events, medians of `--reps`.  GPU box."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="c5:21,c5:31,c5:101,c3:31")
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    import numpy as np
    import torch
    from memo_amd import _lib, synth
    lib = _lib.lib()
    stream = torch.cuda.current_stream()

    def timed(fn):
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for e0, e1 in ev:
            e0.record(stream)
            fn()
            e1.record(stream)
        torch.cuda.synchronize()
        return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))

    for wl in a.workloads.split(","):
        name, k = wl.split(":")
        k = int(k)
        num_docs, L = (500, 1 << 25) if name == "c5" else (100, 100_000_000)
        vb = 2 if num_docs > 255 else 1
        pivot = 8 * L
        ix, (r0, r1) = synth.device_index(3 * L, 4 * L, k, num_docs, pivot, pack="only" if k > 64 else "dense")
        with ix:
            ix.prepare(k, num_docs)
            out = torch.empty(L, dtype=torch.int16 if vb == 2 else torch.uint8, device="cuda:0")
            sweep = (lambda: ix.conservation_dev(3 * L, 4 * L, k, num_docs, out, stream.cuda_stream)) if vb == 2 else \
                    (lambda: ix.conservation_u8_dev(3 * L, 4 * L, k, num_docs, out, stream.cuda_stream))
            for _ in range(100):
                sweep()
            t_sweep = timed(sweep)
            pack = lib.memo_transport_runs16_pack_dev if vb == 2 else lib.memo_transport_runs_pack_dev
            unpack = lib.memo_transport_runs16_unpack_dev if vb == 2 else lib.memo_transport_runs_unpack_dev
            trial = (L * vb + 4 * (L // 32768 + 1) + 3) & ~3
            wire = torch.empty(lib.memo_transport_runs_bytes(L, trial), dtype=torch.uint8, device="cuda:0")
            _lib.check(pack(out.data_ptr(), L, trial, wire.data_ptr(), 0, stream.cuda_stream))
            taken, room = C.c_uint32(), C.c_uint32()
            _lib.check(lib.memo_transport_runs_stats(wire.data_ptr(), 0, stream.cuda_stream, C.byref(taken), C.byref(room)))
            cap = (taken.value + taken.value // 16 + 4096 + 3) & ~3
            wires = [torch.empty(lib.memo_transport_runs_bytes(L, cap), dtype=torch.uint8, device="cuda:0") for _ in range(7)]
            outs = [torch.empty_like(out) for _ in range(7)]
            t_pack = timed(lambda: _lib.check(pack(out.data_ptr(), L, cap, wires[0].data_ptr(), 0, stream.cuda_stream)))
            for w in wires[1:]:
                w.copy_(wires[0])
            row = {"workload": name, "k": k, "positions": L, "value_bytes": vb, "sweep_ms": round(t_sweep, 4), "encode_ms": round(t_pack, 4),
                   "wire_bytes": wires[0].numel(), "bits_per_position": round(8 * wires[0].numel() / L, 3)}
            for cnt in (1, 3, 7):
                ws = (C.c_void_p * cnt)(*[w.data_ptr() for w in wires[:cnt]])
                os_ = (C.c_void_p * cnt)(*[o.data_ptr() for o in outs[:cnt]])
                each = timed(lambda: [_lib.check(unpack(wires[i].data_ptr(), L, cap, outs[i].data_ptr(), 0, stream.cuda_stream)) for i in range(cnt)])
                many = timed(lambda: _lib.check(lib.memo_transport_runs_unpack_many_dev(ws, os_, cnt, L, cap, vb, 0, stream.cuda_stream)))
                row[f"decode_{cnt}_slices_ms"] = {"a_launch_each": round(each, 4), "one_launch": round(many, 4)}
            torch.cuda.synchronize()
            assert all(bool(torch.equal(o, out)) for o in outs), "a decoded slice differs from the result it was coded from"
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
