"""bench.py, N > 1: the gather of the ranks' result slices to rank 0 -- the process group (RCCL; or the gloo test
transport through host memory), the link probe, the choice of a transport coding and of rank 0's share by the step model
(memo_amd.shard), the double-buffered send / receive / decode pipeline of the timed region, the sample of the gathered
result that bench.py checks, and the config-5 leg an N > 1 line carries next to its headline.  The path shards with no data-path
collective (memo_query.py:25-27 with :100 applied to a sub-window): the gather of disjoint slices is the only exchange."""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from memo_amd import _lib, shard, synth


class Transport:
    """The process group and the three collectives the untimed legs use.  TEST transport
    (tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu): RCCL refuses two ranks on one device and this pool gives
    one GPU per box, so the N > 1 control flow -- link probe, coding choice, root weight, per-step send / receive, decode
    on rank 0, parity of the gathered result -- can run here only with every rank on GPU 0 (MEMO_BENCH_ONE_DEVICE=1) and
    gloo carrying the bytes through host memory (MEMO_BENCH_BACKEND=gloo).  The line says so ("test_transport"); its
    numbers are not bench numbers."""

    def __init__(self, rank, world, dev, force_dist):
        self.rank, self.world, self.dev = rank, world, dev
        self.backend = os.environ.get("MEMO_BENCH_BACKEND", "nccl")
        if self.backend not in ("nccl", "gloo"):
            raise SystemExit("MEMO_BENCH_BACKEND: nccl (RCCL; the bench) or gloo (test transport through host memory)")
        self.staged = self.backend == "gloo"
        self.multi = world > 1 or force_dist
        if not self.multi:
            return
        extra = {} if self.staged else {"device_id": dev}
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            dist.init_process_group(self.backend, rank=0, world_size=1, **extra)
        else:
            dist.init_process_group(self.backend, **extra)

    def barrier(self):
        dist.barrier()

    def all_reduce_max(self, x):                  # (device tensor; through host memory on the test transport)
        if not self.staged:
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            return
        h = x.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        x.copy_(h)

    def broadcast0(self, x):
        if not self.staged:
            dist.broadcast(x, src=0)
            return
        h = x.cpu()
        dist.broadcast(h, src=0)
        x.copy_(h)

    def gather0(self, x, bufs):
        if not self.staged:
            dist.gather(x, bufs, dst=0)
            return
        h = x.cpu()
        hb = [torch.empty_like(h) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(h, hb, dst=0)
        if self.rank == 0:
            for d_, s_ in zip(bufs, hb):
                d_.copy_(s_)

    def close(self):
        if self.multi:
            dist.destroy_process_group()


def probe_link(R, T):
    """Who is here, and what a peer's link into rank 0 delivers -- measured, not assumed: a gather of one plain slice per
    rank (what --plain-gather would send every step), 3 repetitions after one untimed.  Every peer sends over its own
    xGMI link at once, so slice bytes / gather time is the per-peer rate the step model needs
    (shard.XGMI_LINK_BYTES_PER_S is only the stand-in for a run with a single rank, where nothing travels).
    Returns (link bytes/s or None, ranks_seen, link_probe)."""
    props = torch.cuda.get_device_properties(R.local)
    me = {"rank": R.rank, "device": R.local, "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None),
          "uuid": str(getattr(props, "uuid", ""))}
    seen = [None] * R.world
    dist.all_gather_object(seen, me)
    ranks_seen = {"world_size": dist.get_world_size(), "ranks": seen,
                  "distinct_devices": len({(r["device"], r["pci_bus_id"], r["uuid"]) for r in seen})}
    R.launch(R.outs[0])
    probe = R.outs[0].view(torch.uint8).reshape(-1)
    probe = probe[:min(probe.numel(), 1 << 28)]
    bufs = [torch.empty_like(probe) for _ in range(R.world)] if R.rank == 0 else None
    T.gather0(probe, bufs)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t_g0 = time.perf_counter()
    for _ in range(3):
        T.gather0(probe, bufs)
    torch.cuda.synchronize()
    t_g = torch.tensor([(time.perf_counter() - t_g0) / 3], dtype=torch.float64, device=R.dev)
    T.all_reduce_max(t_g)
    link_probe = {"bytes_per_rank": probe.numel(), "gather_ms": float(t_g[0]) * 1e3, "reps": 3}
    link_measured = probe.numel() / float(t_g[0]) if R.world > 1 else None
    return link_measured, ranks_seen, link_probe


class Plan:
    """What travels to rank 0, and how much of a share rank 0 sweeps.  A slice's own xGMI link is what bounds N > 1
    (DESIGN.md section 7), so conservation slices go in a lossless transport coding: "runs" (one bit per position + a value
    per change of value, memo_transport_runs_*; carries uint16 slices too), "dense" (2 bits per position + a nibble per
    value outside 1..3, memo_transport_dense_*), "nibble" (one nibble per position, memo_transport_*), or plain bytes.
    Every rank sizes the codings on its own first result (the same query runs every step); a coding is usable when every
    rank's slice fits it.  Fewer bytes is not all: rank 0 decodes world - 1 slices per step, so among the usable codings
    it takes the one whose modelled step
        max(sweep + encode [a peer], sweep + (world - 1) * decode of one slice [rank 0], wire bytes / link [a peer's link])
    (memo_amd.shard.pick_coding / pick_plan) is shortest, with sweep, encode and decode timed here on rank 0.
    Membership slices travel plain (their complement is sparse but no coding for them is built): the line says so."""

    def __init__(self, R, T, link_measured):
        args, lib, L, world, local, dev = R.args, R.lib, R.L, R.world, R.local, R.dev
        stream, outs = R.stream, R.outs
        self.coding, self.b_cap = "plain", 0
        self.cap = max(L // 256, 1024)
        self.choice = None
        self.link = link_measured or shard.XGMI_LINK_BYTES_PER_S
        self.root_weight = 1.0 if args.root_weight == "auto" else min(max(float(args.root_weight), 0.01), 1.0)
        self.vbytes = vbytes = 1 if R.narrow else 2                  # bytes per conservation value
        self.runs_pack_fn = lib.memo_transport_runs_pack_dev if R.narrow else lib.memo_transport_runs16_pack_dev
        self.runs_unpack_fn = lib.memo_transport_runs_unpack_dev if R.narrow else lib.memo_transport_runs16_unpack_dev
        self.t_sweep = None
        self.usable = None
        codable = R.multi and not R.membership and not args.plain_gather
        if not codable:
            return
        found, have, taken, room = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()

        def timed(fn, reps=3):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                fn()
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e-3

        self.t_sweep = t_sweep = timed(lambda: R.launch(outs[0]))
        scratch = torch.empty(L * vbytes, dtype=torch.uint8, device=dev)
        usable = {"plain": (L * vbytes, 0.0, 0.0)}                         # coding -> (wire bytes, decode s, encode s)
        dense_cap = nibble_cap = runs_cap = 0
        if R.narrow:                               # (the dense and nibble codings carry uint8 values only)
            # dense
            trial_b = ((L // 2 + 4 * (L // 32768 + 1)) + 3) & ~3              # every position an escape
            probe = torch.empty(lib.memo_transport_dense_bytes(L, trial_b, self.cap), dtype=torch.uint8, device=dev)
            _lib.check(lib.memo_transport_dense_pack_dev(outs[0].data_ptr(), L, trial_b, self.cap, probe.data_ptr(), local,
                                                         stream.cuda_stream))
            _lib.check(lib.memo_transport_dense_stats(probe.data_ptr(), local, stream.cuda_stream, C.byref(found),
                                                      C.byref(have), C.byref(taken), C.byref(room)))
            need = torch.tensor([taken.value, found.value], dtype=torch.int64, device=dev)
            T.all_reduce_max(need)
            self.b_cap = (int(need[0].item()) + 4096 + 3) & ~3                      # exact + slack
            dense_cap = int(need[1].item()) + 1024
            if not args.nibble_gather:
                t = timed(lambda: _lib.check(lib.memo_transport_dense_unpack_dev(
                    probe.data_ptr(), L, trial_b, self.cap, scratch.data_ptr(), local, stream.cuda_stream)))
                tp = timed(lambda: _lib.check(lib.memo_transport_dense_pack_dev(
                    outs[0].data_ptr(), L, trial_b, self.cap, probe.data_ptr(), local, stream.cuda_stream)))
                usable["dense"] = (lib.memo_transport_dense_bytes(L, self.b_cap, dense_cap), t, tp)
            # nibble
            probe = torch.empty(lib.memo_transport_bytes(L, self.cap), dtype=torch.uint8, device=dev)
            _lib.check(lib.memo_transport_pack_dev(outs[0].data_ptr(), L, self.cap, probe.data_ptr(), local, stream.cuda_stream))
            _lib.check(lib.memo_transport_exceptions(probe.data_ptr(), local, stream.cuda_stream, C.byref(found), C.byref(have)))
            need = torch.tensor([found.value], dtype=torch.int64, device=dev)
            T.all_reduce_max(need)
            nibble_cap = int(need.item()) + int(need.item()) // 16 + 1024          # what the ranks found + slack
            if nibble_cap * 8 <= L // 8:                                       # else the list outweighs the saving
                t = timed(lambda: _lib.check(lib.memo_transport_unpack_dev(
                    probe.data_ptr(), L, scratch.data_ptr(), local, stream.cuda_stream)))
                tp = timed(lambda: _lib.check(lib.memo_transport_pack_dev(
                    outs[0].data_ptr(), L, self.cap, probe.data_ptr(), local, stream.cuda_stream)))
                usable["nibble"] = (lib.memo_transport_bytes(L, nibble_cap), t, tp)
        # runs: one bit per position + one value (one or two bytes: memo_transport_runs16_*) per change of value -- the
        # coding that carries BASELINE config 5's uint16 slices too (67 MB plain against ~9 MB at k = 31)
        trial_r = (L * vbytes + 4 * (L // 32768 + 1) + 3) & ~3            # every position a change
        probe = torch.empty(lib.memo_transport_runs_bytes(L, trial_r), dtype=torch.uint8, device=dev)
        _lib.check(self.runs_pack_fn(outs[0].data_ptr(), L, trial_r, probe.data_ptr(), local, stream.cuda_stream))
        _lib.check(lib.memo_transport_runs_stats(probe.data_ptr(), local, stream.cuda_stream, C.byref(taken), C.byref(room)))
        need = torch.tensor([taken.value], dtype=torch.int64, device=dev)
        T.all_reduce_max(need)
        runs_cap = (int(need.item()) + int(need.item()) // 16 + 4096 + 3) & ~3       # what the ranks needed + slack
        if not args.nibble_gather:
            t = timed(lambda: _lib.check(self.runs_unpack_fn(
                probe.data_ptr(), L, trial_r, scratch.data_ptr(), local, stream.cuda_stream)))
            tp = timed(lambda: _lib.check(self.runs_pack_fn(
                outs[0].data_ptr(), L, trial_r, probe.data_ptr(), local, stream.cuda_stream)))
            # ... and the world - 1 slices of a step by ONE launch (memo_transport_runs_unpack_many_dev: what finish() does)
            t_all = 0.0
            if world > 1:
                many_w = (C.c_void_p * (world - 1))(*([probe.data_ptr()] * (world - 1)))
                many_o = (C.c_void_p * (world - 1))(*([scratch.data_ptr()] * (world - 1)))      # (timing only: one target)
                t_all = timed(lambda: _lib.check(lib.memo_transport_runs_unpack_many_dev(many_w, many_o, world - 1, L, trial_r, vbytes, local,
                                                                                          stream.cuda_stream)))
            usable["runs"] = (lib.memo_transport_runs_bytes(L, runs_cap), t, tp, t_all)
        del probe, scratch
        self.usable = usable
        best, model = shard.pick_coding(world, t_sweep, usable, link=self.link)
        w_best = 1.0
        if args.root_weight == "auto":
            best, w_best, _ = shard.pick_plan(world, t_sweep, usable, link=self.link)
        else:
            w_best = min(max(float(args.root_weight), 0.01), 1.0)
        if args.coding in usable:
            best = args.coding
        names = sorted(usable)
        pick = torch.tensor([names.index(best), int(round(w_best * 1000))], device=dev)
        T.broadcast0(pick)                                        # rank 0's timings decide for everybody
        self.coding = coding = names[int(pick[0].item())]
        self.root_weight = int(pick[1].item()) / 1000.0
        if coding == "dense":
            self.cap = dense_cap
        elif coding == "nibble":
            self.cap = nibble_cap
        elif coding == "runs":
            self.b_cap = runs_cap
        self.choice = {"picked": coding, "root_weight": self.root_weight, "sweep_ms": t_sweep * 1e3,
                       "link_GBs_used": self.link / 1e9,
                       "link_source": "measured in this run (plain gather of one slice per rank)" if link_measured else
                                      "assumed (one rank: nothing travels)",
                       "modelled_step_ms": shard.modelled_step(world, t_sweep, *usable[coding], link=self.link,
                                                               root_weight=self.root_weight) * 1e3,
                       "candidates": {c: {"wire_bytes": int(usable[c][0]), "decode_ms_per_slice": usable[c][1] * 1e3, "encode_ms": usable[c][2] * 1e3,
                                          "decode_ms_all_slices_one_launch": usable[c][3] * 1e3 if len(usable[c]) > 3 else None,
                                          "modelled_step_ms": model[c] * 1e3} for c in names}}

    def expected(self, R, kern_ms):
        """What the step model says THIS run should deliver, from this run's own numbers (the sweep as timed after the
        region, the coding's encode / decode as timed before it, the link as the probe measured it): next to the
        measured `value`, one line says whether the hardware or the model is off."""
        world, L = R.world, R.L
        t_sweep = kern_ms * 1e-3
        if self.usable and self.coding in self.usable:
            u = self.usable[self.coding]
        else:
            u = (int(np.prod(R.shape)) * R.outs[0].element_size(), 0.0, 0.0)     # plain: the result bytes of one slice
        step = shard.modelled_step(world, t_sweep, *u, link=self.link, root_weight=self.root_weight)
        parts = {"peer_sweep_plus_encode_ms": (t_sweep + u[2]) * 1e3,
                 "root_sweep_plus_decode_ms": (self.root_weight * t_sweep + (u[3] if len(u) > 3 and u[3] is not None else (world - 1) * u[1])) * 1e3,
                 "link_ms": u[0] / self.link * 1e3}
        share0 = max(8, int(L * self.root_weight) // 8 * 8) if R.multi else L
        positions = L * (world - 1) + share0
        return {"value": positions / step, "unit": "query-positions/s", "step_ms": step * 1e3, "bound_by": max(parts, key=parts.get),
                **parts, "wire_bytes_per_slice": int(u[0]), "link_GBs": self.link / 1e9, "coding": self.coding,
                "root_weight": self.root_weight,
                "what": "memo_amd.shard.modelled_step with this run's own sweep time, the coding's encode / decode times and the "
                        "link rate of its probe gather"}


class Gather:
    """The timed region's per-step work at N > 1: sweep -> (encode) -> grouped send / recv, one per peer link,
    double-buffered under the next sweep; rank 0 decodes the slices of a step (one launch for the runs coding)."""

    def __init__(self, R, T, P):
        self.R, self.T, self.P = R, T, P
        lib, L, world, dev, nbuf = R.lib, R.L, R.world, R.dev, R.nbuf
        coding, b_cap, cap = P.coding, P.b_cap, P.cap
        self.coded = coded = coding != "plain"
        if coding == "runs":
            self.wires = [torch.empty(lib.memo_transport_runs_bytes(L, b_cap), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        elif coding == "dense":
            self.wires = [torch.empty(lib.memo_transport_dense_bytes(L, b_cap, cap), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        elif coding == "nibble":
            self.wires = [torch.empty(lib.memo_transport_bytes(L, cap), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        else:
            self.wires = [o.view(torch.uint8).reshape(-1) for o in R.outs]  # RCCL has no 16-bit integer type
        self.roots = [[torch.empty_like(self.wires[0]) for _ in range(world)] if (R.multi and R.rank == 0) else None
                      for _ in range(nbuf)]
        if R.multi and R.rank == 0 and coding == "plain":
            for b in range(nbuf):               # plain bytes: rank 0 sweeps straight into its place of the gathered result
                self.roots[b][0] = self.wires[b]
        # rank 0: the gathered slices in result form (decoded when they travelled coded)
        self.decoded = [[torch.empty(L, dtype=R.dtype, device=dev) for _ in range(world)] if (coded and R.rank == 0) else None
                        for _ in range(nbuf)]
        self.pending = [None] * nbuf
        self.host_side = [None] * nbuf               # (test transport only)
        # rank 0's own slice never travels: its sweep writes straight into the gathered result (no pack, no
        # unpack; the wire it contributes to the gather is ignored).  --code-own-slice codes it like a
        # peer's, so that one GPU can exercise the whole path.
        self.skip_own = coded and R.rank == 0 and not R.args.code_own_slice
        if self.skip_own:
            R.outs = [self.decoded[b][0] for b in range(nbuf)]

    def pack(self, src, wire):
        R, P, lib = self.R, self.P, self.R.lib
        if P.coding == "runs":
            _lib.check(P.runs_pack_fn(src.data_ptr(), R.L, P.b_cap, wire.data_ptr(), R.local, R.stream.cuda_stream))
        elif P.coding == "dense":
            _lib.check(lib.memo_transport_dense_pack_dev(src.data_ptr(), R.L, P.b_cap, P.cap, wire.data_ptr(), R.local,
                                                         R.stream.cuda_stream))
        else:
            _lib.check(lib.memo_transport_pack_dev(src.data_ptr(), R.L, P.cap, wire.data_ptr(), R.local, R.stream.cuda_stream))

    def unpack(self, wire, dst):
        R, P, lib = self.R, self.P, self.R.lib
        if P.coding == "runs":
            _lib.check(P.runs_unpack_fn(wire.data_ptr(), R.L, P.b_cap, dst.data_ptr(), R.local, R.stream.cuda_stream))
        elif P.coding == "dense":
            _lib.check(lib.memo_transport_dense_unpack_dev(wire.data_ptr(), R.L, P.b_cap, P.cap, dst.data_ptr(), R.local,
                                                           R.stream.cuda_stream))
        else:
            _lib.check(lib.memo_transport_unpack_dev(wire.data_ptr(), R.L, dst.data_ptr(), R.local, R.stream.cuda_stream))

    def send_slices(self, b):
        """every peer's wire b -> rank 0 (roots[b][g]): grouped RCCL send / recv, one per peer link; rank 0's own slice
        never travels (dist.gather would copy it through RCCL's copy kernel, next to the sweep: 10 % of a step at
        N = 1, gpurun r3a)."""
        R, staged, world = self.R, self.T.staged, self.R.world
        if R.rank == 0:
            if self.coded and R.args.code_own_slice:
                self.roots[b][0].copy_(self.wires[b])     # (validation mode: rank 0's slice coded like a peer's)
            if staged:
                self.host_side[b] = [None] + [torch.empty(self.wires[b].numel(), dtype=torch.uint8) for _ in range(1, world)]
            ops = [dist.P2POp(dist.irecv, self.host_side[b][g] if staged else self.roots[b][g], g) for g in range(1, world)]
        else:
            if staged:
                self.host_side[b] = self.wires[b].cpu()   # (test transport: the wire leaves through host memory)
            ops = [dist.P2POp(dist.isend, self.host_side[b] if staged else self.wires[b], 0)]
        return dist.batch_isend_irecv(ops) if ops else []

    def finish(self, b):                          # gather b done -> (root) slices back in result form
        R, P, world = self.R, self.P, self.R.world
        if self.pending[b] is None:
            return
        for work in self.pending[b]:
            work.wait()
        self.pending[b] = None
        if self.T.staged and R.rank == 0:
            for g in range(1, world):
                self.roots[b][g].copy_(self.host_side[b][g])
        if self.coded and R.rank == 0:
            first = 1 if self.skip_own else 0
            if P.coding == "runs" and world - first > 0:      # every slice of the step by one launch
                cnt = world - first
                ws = (C.c_void_p * cnt)(*[self.roots[b][g].data_ptr() for g in range(first, world)])
                os_ = (C.c_void_p * cnt)(*[self.decoded[b][g].data_ptr() for g in range(first, world)])
                _lib.check(R.lib.memo_transport_runs_unpack_many_dev(ws, os_, cnt, R.L, P.b_cap, P.vbytes, R.local, R.stream.cuda_stream))
            else:
                for g in range(first, world):
                    self.unpack(self.roots[b][g], self.decoded[b][g])

    def step(self, i):
        R = self.R
        b = i % R.nbuf
        self.finish(b)                           # buffer b is free once its previous gather is done
        R.launch(R.outs[b])
        if R.multi:                              # result slices -> rank 0 over xGMI (RCCL send/recv)
            if self.coded and not self.skip_own:
                self.pack(R.outs[b], self.wires[b])
            self.pending[b] = self.send_slices(b)

    def drain(self):
        for b in range(self.R.nbuf):
            self.finish(b)
        torch.cuda.synchronize()

    # ---- after the timed region ---------------------------------------------------------------------------
    def last_rank_sample(self, steps, S=200_000):
        """(rank g, first position a, the first S positions of rank g's gathered slice as the oracle types them, every gathered
        slice of the last step complete | None): what bench.py checks against the oracle -- this module never touches it"""
        R, P, world, L = self.R, self.P, self.R.world, self.R.L
        g = world - 1
        last = (steps - 1) % R.nbuf
        got = (self.decoded[last][g] if self.coded else self.roots[last][g].view(R.dtype).reshape(R.shape))[:S].cpu().numpy()
        got = got.view(np.uint32) if R.membership else got.view(np.uint8 if R.narrow else np.uint16).astype(np.uint16)
        whole = None
        if P.coding != "plain":       # every gathered slice of the last step decodes to the sender's own result?
            whole = True            # (rank 0's own slice against its buffer; the peers' wires report their fill)
            own = self.decoded[last][0]
            whole &= bool(torch.equal(own, R.outs[last].reshape(-1)))
            for g2 in range(1 if self.skip_own else 0, world):
                head = self.roots[last][g2][:16].cpu().numpy().view(np.uint32)
                # head words: runs = (B taken, B capacity, 0, 0); dense = (exceptions, their capacity, B taken, B capacity);
                # nibble = (exceptions, their capacity, overflow flag, -)
                whole &= bool(head[0] <= head[1]) and bool(head[2] <= head[3] if P.coding == "dense" else head[2] == 0)
        return g, g * L, got, whole

    def payload_note(self):
        R, P, L = self.R, self.P, self.R.L
        n = self.wires[0].numel()
        if P.coding == "runs":
            return (f"runs coding: 1 bit per position + {P.b_cap} B for {P.vbytes} byte(s) per change of value "
                    f"({n} B per slice, {8 * n / L:.2f} bits per position)")
        if P.coding == "dense":
            return (f"dense coding: 2 bits per position + {P.b_cap} B of escape nibbles + {P.cap} exception slots "
                    f"({n} B per slice, {8 * n / L:.2f} bits per position)")
        if P.coding == "nibble":
            return f"nibble per position + {P.cap} exception slots ({n} B per slice)"
        note = f"plain result bytes ({n} B per slice)"
        if R.membership:
            note += (f": membership slices have no transport coding -- link-bound: the step model puts a step at "
                     f"{n / P.link * 1e3:.2f} ms on a {P.link / 1e9:.0f} GB/s peer link whatever the sweep takes")
        return note


def config5_leg(args, rank, world, local, T, link_measured, check_sample):
    """First hardware contact must record the configuration the >= 6x target is reachable on: BASELINE's multi-GPU
    configuration is config 5 (500 genomes, 2^25 positions and 8.4e8 rows per GPU, uint16 results, k in {21, 31, 101}),
    not config 3 weak-scaled -- whose sweep is so short (0.17 ms per 10^8 positions) that the 23 MB a peer sends per step
    bound it at ~4x on the step model.  So an N > 1 line whose headline is another workload carries a short config-5 run
    (k = 31, --steps as given, same barrier + max-over-ranks timing) AFTER its timed region, like cpu_baseline at N = 1.
    Partition: memo_query.py:25-27 with :100.  check_sample(R, G, steps): bench.py's check of the gathered result against the
    oracle.  Returns the dict for the line (rank 0) or None."""
    import copy
    from memo_amd import bench_legs
    a5 = copy.copy(args)
    a5.workload, a5.k, a5.rows, a5.wide, a5.rows_file = "c5", 31, "auto", False, None
    a5.headline_only, a5.calibrate, a5.cpu_sample = True, False, 0
    R = bench_legs.Run(a5, rank, world, local)
    try:
        R.build_indexes(T.barrier)
        R.result_layout()
        R.prepare_views()
        P = Plan(R, T, link_measured)
        L = R.L
        share0 = max(8, int(L * P.root_weight) // 8 * 8)
        for o in R.outs:
            R.launch(o)
        if rank == 0:
            R.qe_mine = R.qs + share0
        positions = L * (world - 1) + share0
        G = Gather(R, T, P)
        for i in range(max(2, args.warmup)):
            G.step(i)
        G.drain()
        R.ix.check(R.stream.cuda_stream)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            G.step(i)
        G.drain()
        dt = time.perf_counter() - t0
        dist.barrier()
        R.ix.check(R.stream.cuda_stream)
        per = R.per_step(lambda: R.launch(R.outs[args.steps % R.nbuf]), max(args.steps, 10))
        R.launch(R.outs[args.steps % R.nbuf])
        t = torch.tensor([dt, float(np.mean(per))], dtype=torch.float64, device=R.dev)
        T.all_reduce_max(t)
        dt, kern_ms = float(t[0]), float(t[1])
        if rank != 0:
            return None
        parity = check_sample(R, G, args.steps)
        value = positions * args.steps / dt
        one = L / (kern_ms * 1e-3)              # one GPU sweeping its window, nothing travelling (this run's own sweep time)
        exp = P.expected(R, kern_ms)
        return {"what": "BASELINE config 5 (500 genomes, 2^25 positions and %d rows per GPU, k = 31, uint16 results), run after the "
                        "headline's timed region with the same barriers: the configuration the >= 6x target of 8 GPUs is set on" % R.rows,
                "value": value, "unit": "query-positions/s", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps,
                "kernel_ms": kern_ms, "x_one_gpu": value / one,
                "x_one_gpu_model": exp["value"] / one, "expected_from_model": exp,
                "without_gather": {"value": positions / (kern_ms * 1e-3), "unit": "query-positions/s"},
                "coding": P.coding, "root_weight": P.root_weight, "gather_payload": G.payload_note(),
                "link_GBs_measured": link_measured / 1e9 if link_measured else None,
                "gather_parity_sample": parity}
    finally:
        R.close()
