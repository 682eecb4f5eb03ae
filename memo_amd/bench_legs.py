"""bench.py's untimed legs (N = 1 and N > 1 alike): the state of a run (`Run`), the resident indexes per row format with
their one-time passes timed, the k-class view passes, the same query on the other row formats, the reference's seam
(memo_conservation, host arrays in and out), and the pieces of the JSON line that describe them.
Nothing here is inside bench.py's timed region; bench.py owns argument parsing, the launcher, the timed steps and the two
legs that use the oracle (cpu_baseline, the check of a gathered slice); memo_amd/bench_gather.py the N > 1 gather.  Like
everything under memo_amd/, this module never touches oracle/."""
import json
import os
import sys
import time

import numpy as np
import torch

import memo_amd
from memo_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)

WORKLOADS = {
    # name: (num_docs, window positions per GPU, membership)
    "c2": (10, 10_000_000, False),
    "c3": (100, 100_000_000, False),
    "c4": (100, 100_000_000, True),
    "c5": (500, 1 << 25, False),
    "sparse": (10, 400_000_000, False),      # 0.5 rows per position (tuning experiments)
    "c5h": (250, 1 << 25, False),            # config 5's 25 rows per position with annots that fit a byte (tools/ab.py --density 10/100)
}


class Run:
    """What one rank of one bench run knows.  Filled in the order bench.py calls: Run() -> build_indexes() ->
    result_layout() -> prepare_views() -> (N > 1: bench_gather) -> the legs."""

    def __init__(self, args, rank, world, local):
        self.args, self.rank, self.world, self.local = args, rank, world, local
        self.dev = torch.device("cuda", local)
        self.multi = world > 1 or args.force_dist
        self.lib = _lib.lib()
        self.num_docs, self.L, self.membership = WORKLOADS[args.workload]
        self.host_rows = None
        if args.rows_file:                       # an index built elsewhere (sequences -> matching statistics -> dap_to_bed)
            if world != 1:
                raise SystemExit("--rows-file is a single-GPU workload")
            z = np.load(args.rows_file)
            self.host_rows = tuple(np.ascontiguousarray(z[c], dtype=np.int64) for c in ("start", "end", "annot"))
            self.num_docs, self.L, self.membership = int(z["num_docs"]), int(z["length"]), bool(args.membership)
            args.workload = "rows-file"
        self.k = args.k
        self.pivot = self.L * world
        self.qs, self.qe = rank * self.L, (rank + 1) * self.L
        self.qe_mine = self.qe                   # (N > 1: rank 0 may sweep less than its whole window: bench_gather)
        if args.calibrate:
            _lib.use_ab(True)                    # the PMC calibration kernel lives in libmemo_amd_ab.so only
            self.lib = _lib.lib()
        self._choose_format()
        self.indexes, self.pack_pass, self.dense_pass, self.packed_fmt = {}, None, None, 0
        self.view_pass = self.place_pass = self.unplaced = None
        self.view_bytes, self.full_rows, self.prepared = {}, {}, {}

    # ---- which row format the timed sweep reads ---------------------------------------------------------
    def _choose_format(self):
        args, k, num_docs, membership = self.args, self.k, self.num_docs, self.membership
        # the dense rows answer the unclipped conservation sweep only: k <= 64, num_docs <= 511, >= 1 row per position
        # (membership reads them too when an index holds nothing else, 4 % slower than the 4-byte rows: not a bench format)
        # (256 .. 511 genomes -- BASELINE config 5 -- since round 4: the ninth annot bit in the group's spare byte, uint16 results)
        can_dense = (not membership) and k - 1 <= 63 and ((num_docs <= 255 and not args.wide) or 255 < num_docs <= 511) and num_docs >= 20
        if self.host_rows is not None:           # the library's own rule, on the rows the window sees
            from memo_amd.index import dense_rows_can_answer
            hs, _, ho = self.host_rows
            can_dense = not args.wide and len(hs) > 0 and dense_rows_can_answer(len(hs), int(hs[0]), int(hs[-1]), int(ho.max()), k,
                                                                                 num_docs, membership)
        if args.rows == "auto":     # the fastest format that can answer: dense rows (back to back they are 13 % ahead of the
            args.rows = "wide" if k - 1 > 255 else ("dense" if can_dense else "packed")     # 4-byte rows: DESIGN.md section 7)
        if args.rows == "dense" and not can_dense:
            raise SystemExit("--rows dense answers conservation with k <= 64 and num_docs <= 511 only")
        if k - 1 > 255:
            args.rows = "wide"                  # packed rows answer k <= 256 only
        self.can_dense = can_dense
        self.others = (self.world == 1 and not args.force_dist and not args.headline_only)
        self.formats = [args.rows] + ([f for f in ("wide", "packed", "dense") if f != args.rows and
                                       (f != "dense" or can_dense) and (f == "wide" or k - 1 <= 255)] if self.others else [])
        if args.calibrate and "wide" not in self.formats:
            self.formats.append("wide")         # (the PMC calibration kernel streams the int64 columns: 16 known bytes per row)

    # ---- resident indexes of the same rows, one per row format ------------------------------------------
    def build_indexes(self, barrier=None):
        """The int64 columns as uploaded (24 B/row), the packed query format (memo_index_pack: 4 / 6 B/row, int64 columns
        dropped) and the dense rows (memo_index_pack_dense, everything else dropped).  SURVEY.md 8(d): the passes that
        narrow the rows are timed apart from the query -- on the device (HIP event pair around the annot census and the
        packing kernel inside memo_index_pack, buffers allocated by the first call and reused by the second; an event pair
        on the same stream around memo_index_pack_dense).  barrier: N > 1 on the test transport (the ranks share one GPU
        and build one after the other)."""
        args, k, local = self.args, self.k, self.local
        one_device = os.environ.get("MEMO_BENCH_ONE_DEVICE") == "1" and self.world > 1
        for turn in range(self.world if one_device else 1):
            if one_device:
                barrier()
            if one_device and turn != self.rank:
                continue
            for f in self.formats:
                if self.host_rows is not None:
                    ixf = memo_amd.DeviceIndex.from_host(*self.host_rows, device=local)
                    r0 = int(np.searchsorted(self.host_rows[0], self.qs, side="right"))
                    r1 = int(np.searchsorted(self.host_rows[0], self.qe + k, side="left"))
                else:
                    ixf, (r0, r1) = synth.device_index(self.qs, self.qe, k, self.num_docs, self.pivot, device=local)
                nrows = r1 - r0
                if f != "wide":
                    ixf.pack(keep_wide=True)            # allocates the packed rows
                    ixf.pack(keep_wide=True)            # the timed pass: same buffers
                    info = ixf.info()
                    self.packed_fmt = info["packed_format"]
                    pk_bytes = 6 if self.packed_fmt == 6 else 4
                    pack_bytes = (24 + pk_bytes) * nrows + 2 * pk_bytes * nrows        # + the order inside the buckets: the words read and written again
                    self.pack_pass = {
                        "what": "memo_index_pack: int64 columns -> packed rows, once per index, not part of a query "
                                "(the layout guessed from a sample of the annot column; pack_rows_kernel reads 24 B, writes "
                                "%d B per row (format %d) and takes the exact annot census on the way; then the rows' order inside "
                                "their buckets, in place: interleave_small_kernel)" % (pk_bytes, self.packed_fmt),
                        "ms": info["pack_ms"], "rows": nrows, "bytes": pack_bytes,
                        "GBs": pack_bytes / (info["pack_ms"] * 1e-3) / 1e9 if info["pack_ms"] > 0 else None,
                        "timed_with": "HIP event pair inside memo_index_pack (memo_index_info.pack_ms), second call "
                                      "on reused buffers"}
                    ixf.pack(keep_wide=False)           # drop the int64 columns
                if f == "dense":
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    null_stream = torch.cuda.default_stream()           # memo_index_pack_dense works on the NULL stream
                    e0.record(null_stream)
                    ixf.pack_dense(keep_packed=False)
                    e1.record(null_stream)
                    torch.cuda.synchronize()
                    dms = e0.elapsed_time(e1)
                    self.dense_pass = {
                        "what": "memo_index_pack_dense: 4-byte rows -> dense rows, five per 16 bytes (reads 4 B, writes "
                                "3.2 B per row; includes the hipMalloc of the dense rows), once per index",
                        "ms": dms, "rows": nrows, "bytes": 7.2 * nrows, "GBs": 7.2 * nrows / (dms * 1e-3) / 1e9}
                self.indexes[f] = ixf
        self.rows = r1 - r0
        self.rows_read = {f: self.rows for f in self.indexes}
        self.ix = self.indexes[args.rows]
        self.fmt_bytes = {"wide": 24, "packed": 6 if self.packed_fmt == 6 else 4, "dense": 3.2}
        self.row_bytes = self.fmt_bytes[args.rows]

    # ---- the result: element type, buffers, the launch ---------------------------------------------------
    def result_layout(self):
        L, num_docs = self.L, self.num_docs
        self.W = (num_docs + 31) // 32
        # result element: membership = W uint32 words; conservation = uint8 when num_docs <= 255 (the
        # same at every N; it also halves what the slices put on xGMI), else uint16
        self.narrow = (not self.membership) and num_docs <= 255 and not self.args.wide
        if self.membership:
            self.shape, self.dtype, self.b_out = (L, self.W), torch.int32, 4 * self.W
        elif self.narrow:
            self.shape, self.dtype, self.b_out = (L,), torch.uint8, 1
        else:
            self.shape, self.dtype, self.b_out = (L,), torch.int16, 2          # uint16 payload
        # two result buffers at N > 1: the gather of step i (RCCL stream) overlaps the sweep of step i+1
        self.nbuf = 2 if self.multi else 1
        self.outs = [torch.empty(self.shape, dtype=self.dtype, device=self.dev) for _ in range(self.nbuf)]
        self.stream = torch.cuda.current_stream()

    def launch(self, out, ix=None):
        ix = ix or self.ix
        st = self.stream.cuda_stream
        if self.membership:
            ix.membership_dev(self.qs, self.qe_mine, self.k, self.num_docs, out, st)
        elif self.narrow:
            ix.conservation_u8_dev(self.qs, self.qe_mine, self.k, self.num_docs, out, st)
        else:
            ix.conservation_dev(self.qs, self.qe_mine, self.k, self.num_docs, out, st)

    def per_step(self, fn, n):
        """n launches with a HIP event pair each -> ms per launch (list)"""
        ev_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a_, b_ in ev_:
            a_.record(self.stream)
            fn()
            b_.record(self.stream)
        torch.cuda.synchronize()
        return [a_.elapsed_time(b_) for a_, b_ in ev_]

    def settle(self, fn):
        """untimed launches until the device has settled on this kernel -- a change of kernels sets off a swing of the
        clocks (the sweep runs at the 1400 W power cap: fast for a few launches, then slower than its steady state,
        then back over some thirty launches; profiles/r02_dense_rows_ab.txt): batches of 20 until two in a row are
        within 1 % of the one before them; at least 5 batches, at most 1.5 s of them"""
        prev, calm, n, t_ramp = None, 0, 0, time.perf_counter()
        while True:
            cur = float(np.mean(self.per_step(fn, 20)))
            n += 20
            calm = calm + 1 if prev is not None and abs(cur - prev) <= 0.01 * prev else 0
            prev = cur
            if (calm >= 2 and n >= 100) or time.perf_counter() - t_ramp > 1.5:
                return {"launches": n, "ms": (time.perf_counter() - t_ramp) * 1e3}

    # ---- the k-class views, their passes timed ------------------------------------------------------------
    def prepare_views(self):
        """One launch per resident format before anything is timed: the first conservation query of a k class on the dense
        rows builds their k-class VIEW (the rows whose overlap is below the class's cap -- 2, 4, 6 ... 32 --: all that can
        write at this k; memo_query.py:49 drops the others per query, the library once per index and class --
        memo_index_info_t.last_rows_read) and the tile table; both stay with the index.  What each format's sweep reads is
        what its roofline is priced on."""
        args, k, num_docs, membership = self.args, self.k, self.num_docs, self.membership
        for f, ixf in self.indexes.items():
            if f != "wide":
                # memo_index_prepare: the k-class view and the tile table NOW (a host that sweeps one k over many windows calls it
                # once; without it the view is built by the query that finds it has become worth its pass: MEMO_OPT_BUILD_COST_PCT)
                # -- timed on the device by the library.  Dense rows: in the two steps a sequence of queries would take -- first the
                # view with its rows in the order they come (MEMO_OPT_VIEW_PLACES 0), then again with the rows' places inside their
                # groups chosen against LDS bank conflicts -- each timed, and the sweep on the first timed too.
                if f == "dense" and f == args.rows and not membership:
                    ixf.set_option(5, 0)
                self.prepared[f] = ixf.prepare(k, num_docs, membership)
                inf = ixf.info()
                if f == args.rows and inf["last_view_ms"] > 0:
                    self.view_pass = {
                        "what": "k-class view of the rows the sweep reads (rows whose overlap is below the class's cap: all that can "
                                "write at this k), built by memo_index_prepare (else: by the query that finds the class's queries "
                                "have lost more to its absence than it costs) -- count, scan, one fused pass over "
                                "the rows -- once per index and class, kept",
                        "ms": float(inf["last_view_ms"]), "rows_in": int(inf["dense_row_count"] if f == "dense" else inf["rows"]),
                        "rows_kept": int(inf["last_rows_read"]), "device_bytes_taken_by_prepare": int(self.prepared[f])}
                    # ... and once more on memory the process has held before, like row_format_pass's second call: the pass is timed with
                    # its allocations, and a first hipMalloc of a fresh 0.85 GB region has taken the driver 350 ms on one box of the pool
                    # (gpurun r5valid) where the pass takes 1.7
                    ixf.set_option(1, 0)
                    ixf.set_option(1, 1)
                    ixf.check()
                    ixf.prepare(k, num_docs, membership)
                    again = float(ixf.info()["last_view_ms"])
                    if again > 0:
                        self.view_pass.update({
                            "ms_first_build": self.view_pass["ms"], "ms_second_build": again, "ms": min(self.view_pass["ms"], again),
                            "timed_with": "HIP event pair around the whole pass inside memo_index_prepare (its allocations and two host "
                                          "waits included); built twice, the views dropped in between: `ms` is the smaller"})
                if f == "dense" and f == args.rows and not membership:
                    ixf.set_option(5, 1)
                    if self.view_pass:
                        if self.others:
                            # the sweep on the view as a sequence of queries first gets it (300 launches, the last 100 timed)
                            ms_u = sorted(self.per_step(lambda: self.launch(self.outs[0], ixf), 300)[200:])
                            self.unplaced = {"kernel_ms_median": float(ms_u[len(ms_u) // 2]), "rows_read": int(ixf.info()["last_rows_read"])}
                        ixf.prepare(k, num_docs, membership)
                        inf = ixf.info()
                        if inf["last_view_ms"] > 0:
                            self.place_pass = {
                                "what": "the same view built again with the place of every row inside its 16-byte group chosen against LDS "
                                        "bank conflicts (memo_view.hip: view_place_bucket) -- by memo_index_prepare at once, by queries "
                                        "once the class's queries have lost to the plain view what this pass costs (MEMO_OPT_VIEW_PLACES)",
                                "ms": float(inf["last_view_ms"])}
            self.launch(self.outs[0], ixf)
            torch.cuda.synchronize()
            inf = ixf.info()
            self.full_rows[f] = int(inf["dense_row_count"]) if f == "dense" else self.rows
            if inf["last_rows_read"]:
                self.rows_read[f] = int(inf["last_rows_read"])
            if f == "dense" and inf.get("last_view_rows_per_group") == 6:
                self.view_bytes[f] = 16.0 / 6.0               # a view of six rows per group: bytes per row (slot) the sweep reads
                if f == args.rows:
                    self.row_bytes = self.view_bytes[f]

    # ---- which kernel a format's sweep is ------------------------------------------------------------------
    def kernel_name(self, which):
        membership, packed_fmt = self.membership, self.packed_fmt
        if which == "dense" and not membership:
            inf = self.indexes[which].info()
            if inf["last_sweep"] == 5:
                if inf["last_variant"] == 3:
                    return ("sweep_conservation_halo3t_kernel<..., SIX> (a k-class view of the dense rows as groups of six rows that carry their "
                            "bucket: 2.67 B per row; the tile's row slice from the index's tile table)")
                return ("sweep_conservation_halo3t_kernel<...> (dense rows, five per 16 bytes; the tile's row slice from the index's tile table)"
                        if inf["last_variant"] == 2 else "sweep_conservation_halo3_kernel<...> (PackedRows3: five rows per 16 bytes)")
            return {1: "sweep_conservation_kernel<", 2: "sweep_conservation_halo_kernel<"}.get(inf["last_sweep"], "?<") + "...> (the dense rows could not answer)"
        rows_t = "WideRows" if which == "wide" else ("PackedRows<true, false>" if packed_fmt == 6 else
                                                     "PackedRows<false, false, true>" if packed_fmt == 12 else "PackedRows<false, false>")
        if membership:      # packed rows: per-genome bit planes (unclipped + staged up to 512 genomes,
            if which == "dense":
                return "sweep_membership_planes3_kernel<...> (PackedRows3: five rows per 16 bytes)"
            if which == "wide":                                      # else the clipped "runs"); int64 rows: doubling
                return "sweep_membership_kernel<" + rows_t + ", ...>"
            return ("sweep_membership_planes_kernel<" if self.num_docs <= 512 else "sweep_membership_runs_kernel<") + rows_t + ", ...>"
        # the library says which kernel family answered the last conservation query on this index (it picks the
        # level arrays per query from k and the overlap lengths of the rows: memo_sweep_cons.hip, pick_levels)
        family = {1: "sweep_conservation_kernel<", 2: "sweep_conservation_halo_kernel<", 3: "sweep_conservation_r4_kernel<",
                  4: "sweep_conservation_mixed_kernel<"}[self.indexes[which].info()["last_sweep"]]
        return family + rows_t + ", ...>"

    # ---- the same query on the OTHER row formats, for the record -------------------------------------------
    def other_formats(self):
        """... and first, so that the headline's timed region does not sit in the clock ramp of a cold device (the driver
        runs --steps 20 --warmup 5)."""
        args, k, num_docs, membership, L = self.args, self.k, self.num_docs, self.membership, self.L
        indexes, others = self.indexes, self.others
        other = []
        legs = [(w, w, None) for w in self.formats[1:]]
        if others and args.rows != "wide" and self.rows_read[args.rows] != self.full_rows[args.rows]:
            # the same kernel on ALL the rows of the headline format (MEMO_OPT_VIEWS off: no k-class view), for the record: what round 2 timed
            legs.append((args.rows, args.rows + ", all rows (no k-class view)", self.full_rows[args.rows]))
        if others and args.rows == "dense" and self.view_bytes.get("dense"):
            # the view of FIVE rows per group (MEMO_OPT_VIEW_ROWS 5: rounds 3-4's headline; 3.2 B per row, no padding), for the record
            legs.append(("dense", "dense, the k-class view as groups of five rows (MEMO_OPT_VIEW_ROWS 5)", -5))
        if others and not membership and self.narrow:
            legs.append((args.rows, args.rows + ", uint16 results (memo_query_conservation_dev: the ABI's primary result type)", -16))
        for which, label, all_rows in legs:
            ob = self.fmt_bytes[which] if all_rows and all_rows > 0 else self.view_bytes.get(which, self.fmt_bytes[which])
            fn2, out2, b_out2, read2 = (lambda: self.launch(self.outs[0], indexes[which])), None, self.b_out, None
            if all_rows and all_rows > 0:
                indexes[which].set_option(1, 0)         # MEMO_OPT_VIEWS = 0: the views go, every sweep reads all the rows
            elif all_rows == -5:
                indexes[which].set_option(1, 0)         # (one view resident at a time, as in the headline's own leg)
                indexes[which].set_option(1, 1)
                indexes[which].set_option(4, 5)
                indexes[which].prepare(k, num_docs, membership)
                ob = self.fmt_bytes[which]
            elif all_rows == -16:
                out2 = torch.empty(L, dtype=torch.int16, device=self.dev)
                b_out2 = 2
                fn2 = lambda: indexes[which].conservation_dev(self.qs, self.qe_mine, k, num_docs, out2, self.stream.cuda_stream)
            try:
                self.settle(fn2)
                ms2 = self.per_step(fn2, max(args.steps, 20))
                indexes[which].check(self.stream.cuda_stream)
                name2 = self.kernel_name(which)
                read2 = int(indexes[which].info()["last_rows_read"]) or None
            finally:
                if all_rows and all_rows > 0:
                    indexes[which].set_option(1, 1)
                    indexes[which].prepare(k, num_docs, membership)     # (the headline's view again)
                elif all_rows == -5:
                    indexes[which].set_option(1, 0)
                    indexes[which].set_option(1, 1)
                    indexes[which].set_option(4, 0)
                    indexes[which].prepare(k, num_docs, membership)
            if all_rows and all_rows < 0:
                all_rows = read2
            alg2 = ob * (all_rows or self.rows_read[which]) + b_out2 * L
            med2 = float(np.median(ms2))
            other.append({"rows": label, "row_bytes": ob, "rows_read": all_rows or self.rows_read[which], "kernel": name2,
                          "kernel_ms": float(np.mean(ms2)),
                          "kernel_ms_median": med2, "kernel_ms_min": float(np.min(ms2)),
                          "query_positions_per_s": L / (med2 * 1e-3), "algorithmic_bytes": alg2,
                          "achieved_GBs": alg2 / (med2 * 1e-3) / 1e9, "frac": alg2 / (med2 * 1e-3) / 1e9 / HBM_PEAK_GBS})
        return other

    def result_as_host(self, out, S=None):
        """the device result (first S positions) in the oracle's element type"""
        h = (out if S is None else out[:S]).cpu().numpy()
        if self.membership:
            return h.view(np.uint32)
        return h.view(np.uint8).astype(np.uint16) if self.narrow else h.view(np.uint16)

    def close(self):
        for x in self.indexes.values():
            x.close()


# ---- pieces of the JSON line ----------------------------------------------------------------------------------
def config_entry(R, positions_per_step, root_weight, ramp):
    args, k, num_docs, L, rows, membership, world = R.args, R.k, R.num_docs, R.L, R.rows, R.membership, R.world
    rows_read, view_bytes, full_rows, packed_fmt = R.rows_read, R.view_bytes, R.full_rows, R.packed_fmt
    return {"workload": (f"BASELINE config {args.workload[1:]}: synthetic {num_docs}-genome index, "
                         if R.host_rows is None else
                         f"index from sequences ({os.path.basename(args.rows_file)}: tools/realistic_index.py), {num_docs} genomes, ") +
                        f"{L} positions/GPU window, {rows} rows/GPU, k={k}, "
                        f"{'membership' if membership else 'conservation'}",
            "num_docs": num_docs, "window_per_gpu": L, "rows_per_gpu": rows, "k": k,
            "query": "membership" if membership else "conservation",
            "row_format": "3 x int64 per row as uploaded (24 B)" if args.rows == "wide" else
                          f"packed {6 if packed_fmt == 6 else 4} B/row (format {packed_fmt}) built once per index by memo_index_pack" if args.rows == "packed"
                          else "3.2 B/row (five 24-bit rows per 16 bytes: start mod 2^10, length saturated at 63, 8-bit "
                               "order) built once per index by memo_index_pack + memo_index_pack_dense" +
                               ("" if rows_read.get("dense", rows) == rows else
                                f"; the sweep reads the k-class view of them: the rows whose overlap is below "
                                f"{2 * (k // 2)} -- the others cannot write at k = {k} "
                                "(memo_query.py:49 drops them per query) -- " +
                                (f"as groups of SIX rows that carry their bucket (16 B per 6 rows; every bucket padded to whole groups with "
                                 f"copies of one of its rows: {rows_read['dense']} places in all), " if view_bytes.get("dense") else
                                 f"{rows_read['dense']} rows, ") +
                                "built once per index and class by memo_index_prepare (else by the query that finds the view has become "
                                "worth its pass), timed in dense_view_pass / dense_view_place_pass; `roofline` is priced on the bytes read; "
                                "other_row_formats has the same kernel on all the dense rows and on the view of five rows per group"),
            "row_bytes": R.row_bytes, "rows_read": rows_read[args.rows],
            "rows_read_note": None if args.rows == "wide" or rows_read[args.rows] == full_rows.get(args.rows) else
                              f"the sweep reads the k-class view of the {args.rows} rows: the {rows_read[args.rows]} of "
                              f"{full_rows.get(args.rows)} rows that can write at k = {k} (memo_query.py:49 drops the others per "
                              "query; the library once per index and class -- memo_index_prepare, else the query that finds the view worth its pass: dense_view_pass); "
                              "`roofline` is priced on the rows read, other_row_formats has the same kernel on all the rows",
            "result_bytes_per_position": R.b_out,
            "row_format_choice": "--rows auto = the format that answers this query fastest: the dense rows where they "
                                 "can (conservation, k <= 64, num_docs <= 511), else the 4- / 6-byte rows (k <= 256), "
                                 "else int64.  roofline is priced on the bytes of the format read (fewer bytes per row "
                                 "lower `frac` at the same speed); the other resident formats are timed in "
                                 "other_row_formats",
            "row_format_pass": R.pack_pass, "dense_format_pass": R.dense_pass, "dense_view_pass": R.view_pass,
            "dense_view_place_pass": R.place_pass,
            "clock_ramp": {"what": "untimed headline launches before the warm-up steps, in batches of 20 until two "
                                   "batches in a row are within 1 % of the one before them", **ramp},
            "sharding": f"window split into {world} contiguous sub-windows, slices gathered to rank 0 "
                        f"over RCCL (double-buffered: gather i overlaps sweep i+1); rank 0 sweeps "
                        f"{root_weight:g} of a share ({positions_per_step} positions per step in all)"
                        if world > 1 else "single GPU"}


def roofline_entry(R, kern_ms, per, b_alg):
    achieved = b_alg / (kern_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
            "kernel": R.kernel_name(R.args.rows),
            "kernel_ms": kern_ms, "kernel_ms_median": float(np.median(per)),
            "kernel_ms_min": float(np.min(per)), "algorithmic_bytes": b_alg,
            "timed_with": "HIP events on the launch stream: one pair around the K timed steps (kernel_ms "
                          "= mean); median / min from K further steps with a pair each" if not R.multi else
                          "HIP events on the launch stream: K sweeps with a pair each, after the timed "
                          "region (kernel_ms = mean, slowest rank)"}


def attach_traffic(R, res, b_alg):
    """roofline.traffic from profiles/traffic.json: PMC passes are separate runs (tools/pmc_summary.py)"""
    args, k = R.args, R.k
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(prof):
        return
    key = f"{args.workload}_{args.rows}" if R.host_rows is None else \
        f"{os.path.basename(os.path.dirname(os.path.abspath(args.rows_file)))}_{'memb' if R.membership else 'cons'}_k{k}_{args.rows}"
    allt = json.load(open(prof))
    tj = allt.get(f"{key}_k{k}") or allt.get(key)            # (entries of a k other than the workload's default carry it in their key)
    if tj and tj.get("result_bytes_per_position") == R.b_out and tj.get("algorithmic_bytes") is not None and \
            abs(tj["algorithmic_bytes"] - b_alg) <= 0.01 * b_alg:   # same kernel instantiation, same rows read (an entry that
                                                                    # does not say what it measured matches nothing)
        res["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
        res["roofline"]["traffic_source"] = (
            "profiles/traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
            "(%s), not measured in this run" % tj.get("source", "tools/pmc_summary.py"))
    elif tj:                                                    # (a stale entry must not vanish silently)
        sys.stderr.write(f"bench.py: profiles/traffic.json[{key}] is for {tj.get('algorithmic_bytes')} algorithmic bytes / "
                         f"{tj.get('result_bytes_per_position')} B results, this run has {b_alg} / {R.b_out}: roofline.traffic left empty\n")


def without_view_entry(R, other, kern_ms):
    """what the same resident index delivers BEFORE its k-class view exists, next to `value`"""
    allr = [o for o in other if "all rows" in o["rows"]]
    if not (allr and R.view_pass):
        return None
    view_pass, place_pass, unplaced = R.view_pass, R.place_pass, R.unplaced
    first_ms = unplaced["kernel_ms_median"] if unplaced else kern_ms     # the sweep on the view as queries first get it
    gain_ms = allr[0]["kernel_ms_median"] - first_ms
    return {"value": allr[0]["query_positions_per_s"], "unit": "query-positions/s", "kernel_ms_median": allr[0]["kernel_ms_median"],
            "frac": allr[0]["frac"], "rows_read": allr[0]["rows_read"],
            "view_build_ms": view_pass["ms"],
            "view_amortised_after_queries": (view_pass["ms"] / gain_ms) if gain_ms > 0 else None,
            "sweep_on_the_view_before_its_rows_are_placed": unplaced,
            "places_build_ms": place_pass["ms"] if place_pass else None,
            "places_amortised_after_queries": (place_pass["ms"] / (first_ms - kern_ms)) if (place_pass and unplaced and first_ms > kern_ms) else None,
            "note": "`value` holds for a resident index once memo_index_prepare has built the class's view of the rows with the rows' "
                    "places chosen.  A sequence of queries gets there in two steps, each taken by the query that finds the step has "
                    "paid for itself (ski rental: MEMO_OPT_BUILD_COST_PCT): first the view (view_build_ms, after "
                    "view_amortised_after_queries whole-window queries of the class: until then this -- the same kernel on all the "
                    "rows of the same format, also what every query of an index that answers one gets: the one-shot forms, `memo "
                    "query`), then the same view with its rows placed (places_build_ms, places_amortised_after_queries)"}


# ---- the reference's seam ---------------------------------------------------------------------------------------
def cgroup_throttled_us():
    """microseconds this process's cgroup has spent frozen by its CFS quota so far (cgroup v2 cpu.stat), or None"""
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("throttled_usec"):
                return int(ln.split()[1])
    except (OSError, ValueError, IndexError):
        pass
    return None


def pcie_rates(R, nbytes=256 << 20):
    """What this box's link gives a pinned buffer, both ways (HIP events around one copy of 256 MiB after an untimed
    one): the seam can be no faster than its packed rows over this -- boxes of the pool differ by a factor of three."""
    try:
        host = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        devb = torch.empty(nbytes, dtype=torch.uint8, device=R.dev)
        out = {}
        for name, (dst, src) in (("h2d_GBs", (devb, host)), ("d2h_GBs", (host, devb))):
            dst.copy_(src, non_blocking=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(R.stream)
            dst.copy_(src, non_blocking=True)
            e1.record(R.stream)
            torch.cuda.synchronize()
            out[name] = nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9
        out["what"] = "one copy of 256 MiB between pinned host memory and the device, HIP events"
        return out
    except Exception as exc:
        return {"error": repr(exc)}


def one_shot_seam(R, out):
    """THE REFERENCE'S SEAM (memo_query.py:103-105: memo_init + memo_query + the argmax of print_res): memo_conservation --
    three host int64 columns in, a host uint16 result out; what a maintainer who applies INTEGRATION.md section 2 gets.
    PCIe and the host's packing pass inclusive: never `value`.  The columns come back from the resident int64 index
    (12 GB for config 3).  One call that is not counted (it starts the worker pool and allocates the pinned ring: reported
    as first_call_ms), then three calls on columns whose pages were first touched by the library's own pool threads (what
    the download gives), then three on copies every page of which ONE thread touched first (what a NumPy caller has)."""
    import ctypes as C
    lib, k, num_docs, L, rows = R.lib, R.k, R.num_docs, R.L, R.rows
    try:
        ds, de, do = R.indexes["wide"].columns()
        cols = [np.empty(rows, np.int64) for _ in range(3)]
        for h_, d_ in zip(cols, (ds, de, do)):
            _lib.check(lib.memo_dev_download(R.local, h_.ctypes.data, d_, h_.nbytes, None))

        def calls(c, n):
            ms, seam = [], None
            for _ in range(n):
                t_s = time.perf_counter()
                seam = memo_amd.conservation(c[0], c[1], c[2], R.qs, R.qe, k, num_docs, device=R.local)
                ms.append((time.perf_counter() - t_s) * 1e3)
            return ms, seam
        pcie = pcie_rates(R)
        thr0 = cgroup_throttled_us()
        first_ms, _ = calls(cols, 1)
        ms_pool, seam = calls(cols, 3)
        same = bool(np.array_equal(seam, R.result_as_host(out)))
        one = [c.copy() for c in cols]           # NumPy's copy: every page first written by this thread
        del cols
        ms_one, seam1 = calls(one, 3)
        same1 = bool(np.array_equal(seam1, seam))
        # ... and as the stub of INTEGRATION.md section 2 calls it: filter_pq's own [M, 3] array, row-major, as it lies
        # (memo_conservation_rows; the array is made here from the columns -- NumPy, strided, not timed)
        aos = np.empty((rows, 3), np.int64)
        for j_ in range(3):
            aos[:, j_] = one[j_]
        ms_rows, seam_r = [], None
        for _ in range(3):
            t_s = time.perf_counter()
            seam_r = memo_amd.conservation_rows(aos, R.qs, R.qe, k, num_docs, device=R.local)
            ms_rows.append((time.perf_counter() - t_s) * 1e3)
        same_r = bool(np.array_equal(seam_r, seam))
        del aos, seam_r
        thr1 = cgroup_throttled_us()
        from memo_amd.index import dense_rows_can_answer
        dense_in = dense_rows_can_answer(rows, int(one[0][0]), int(one[0][-1]), 0, k, num_docs, False)
        allowed, quota = C.c_int32(), C.c_double()
        threads = lib.memo_host_threads(C.byref(allowed), C.byref(quota))
        ms = float(np.median(ms_pool))
        bytes_up = (16.0 / 5.0 if dense_in else 4.0) * rows        # what the packed rows put on PCIe
        return {"what": "memo_conservation (include/memo_amd.h): host int64 columns in, host uint16 result out -- the three calls "
                        "memo_query.py:103-105 make; rows packed on the host into pinned memory, PCIe, sweep, result back; the "
                        "index is built and destroyed inside the call",
                "ms": ms, "ms_calls": [float(x) for x in ms_pool], "value": L / (ms * 1e-3),
                "unit": "query-positions/s", "rows": rows, "host_bytes_in": 24 * rows, "host_bytes_out": 2 * L,
                "first_call_ms": float(first_ms[0]),
                "columns_first_touched_by_one_thread": {"ms": float(np.median(ms_one)), "ms_calls": [float(x) for x in ms_one],
                                                        "value": L / (float(np.median(ms_one)) * 1e-3), "parity": same1},
                "reference_array_form": {"what": "memo_conservation_rows: filter_pq's uint64 [M, 3] array, row-major, as it lies -- what the "
                                                 "stub of INTEGRATION.md section 2 calls",
                                         "ms": float(np.median(ms_rows)), "ms_calls": [float(x) for x in ms_rows],
                                         "value": L / (float(np.median(ms_rows)) * 1e-3), "parity": same_r},
                "box_note": "the host is shared by other tenants: the same call has taken 37 ... 113 ms across the pool's boxes and minutes with "
                            "the link at 57 GB/s throughout (profiles/r06_oneshot.txt); pcie = this box's link now, "
                            "cgroup_throttled_ms_during_the_calls = time the container was frozen by its CPU quota while the ten calls ran",
                "note": "`ms` = median of three calls after one uncounted call (first_call_ms: it starts the worker pool and allocates "
                        "the pinned ring); the columns' pages were first touched by the library's pool threads (the download that "
                        "produced them) -- columns_first_touched_by_one_thread: the same on NumPy copies made by one thread",
                "way_in": ("host packer -> dense rows (3.2 B per row over PCIe)" if dense_in else "host packer -> 4-byte words"),
                "pcie": pcie,
                "cgroup_throttled_ms_during_the_calls": (thr1 - thr0) / 1e3 if (thr0 is not None and thr1 is not None) else None,
                "floor_ms_from_pcie": (bytes_up / (pcie["h2d_GBs"] * 1e9) + 2 * L / (pcie["d2h_GBs"] * 1e9)) * 1e3 if pcie.get("h2d_GBs") else None,
                "host_threads": int(threads), "host_cpus_allowed": int(allowed.value),
                "host_cgroup_cpu_quota": float(quota.value) or None,
                "host_cpus_visible": os.cpu_count(), "parity": same}
    except Exception as exc:                         # (the seam's line must not cost the run its headline)
        return {"error": repr(exc)}
