#!/usr/bin/env python3
"""bench.py -- query-positions/sec of the windowed k-mer conservation sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 either way: `python bench.py --gpus N ...` starts its own ranks -- one child
     `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, whose one JSON line and exit
     code it relays -- or start it under torch.distributed.run yourself; see launch_ranks())

Workload (BASELINE.json config 3 by default): synthetic pangenome, 100 genomes, 5 rows per
pivot position (density 0.05 per genome-position), one window of 10^8 positions per GPU,
k = 31, conservation.  A "step" is one query of the whole window: one launch of
sweep_conservation_kernel over the rows resident in HBM, result left in HBM.  Weak scaling:
every rank owns its own 10^8-position window of an N x 10^8 pivot and generates exactly the
rows that window sees (index-addressable generator, memo_amd/synth.py); for N > 1 the result
slices are gathered to rank 0 over RCCL inside the timed region (as bytes; two result buffers so
that gather i overlaps sweep i+1).  Conservation results are uint8 per position when num_docs <= 255
(uint16 otherwise, or with --wide), at every N.

One JSON line on stdout (rank 0).  `roofline` prices the sweep kernel alone from HIP events
recorded on the launch stream; `cpu_baseline` is the oracle's literal port of the
reference loop (memo_query.py:45-63,70) on one host core over a bounded sample window.

Order of a run (N = 1): build both indexes (int64 columns; packed rows, the packing pass timed on the
device by the library) -> the same query on the OTHER row format (reported as other_row_format; real
work that also brings the clocks up) -> clock ramp: headline launches until their time stops falling
(untimed, bounded) -> W warm-up steps -> K timed steps -> K more steps with an event pair each (per-step
min / median, reported next to the mean) -> CPU baseline + parity of the timed result.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# numpy / torch / memo_amd are imported in main(), after launch_ranks() has had its say: the parent of a
# self-launched N > 1 run never loads the HIP runtime or the library.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--wide", action="store_true", help="uint16 conservation results even when num_docs <= 255")
    ap.add_argument("--rows", default="auto", choices=["auto", "dense", "packed", "wide"],
                    help="row format the timed sweep reads: dense (memo_index_pack_dense, 3.2 B/row: conservation, "
                         "k <= 64, num_docs <= 511), packed (memo_index_pack, 4-6 B/row), or the int64 columns as "
                         "uploaded (24 B/row); auto = the fastest that can answer (dense where they can, else packed for k <= 256); at N=1 the "
                         "others are timed too")
    ap.add_argument("--plain-gather", action="store_true",
                    help="N > 1: send uint8 slices as they are (default: the densest transport coding that fits)")
    ap.add_argument("--coding", default=None, choices=["runs", "dense", "nibble", "plain"],
                    help="N > 1: this transport coding instead of the one the step model picks (if the slices fit it)")
    ap.add_argument("--code-own-slice", action="store_true",
                    help="N > 1: rank 0 packs and unpacks its own slice too (exercises the coded path on one GPU)")
    ap.add_argument("--nibble-gather", action="store_true",
                    help="N > 1: skip the dense transport coding (nibble coding when it fits, else plain)")
    ap.add_argument("--root-weight", default="auto",
                    help="N > 1: the share of a window rank 0 sweeps (it also decodes the peers' slices): a number in "
                         "(0, 1], or auto = what the step model says (1 when the slices travel as plain bytes)")
    ap.add_argument("--calibrate", action="store_true",
                    help="also run the read-once calibration kernel (for the FETCH_SIZE PMC pass)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the gather path even with one rank (validation)")
    ap.add_argument("--rows-file", default=None,
                    help="an index from a file instead of the synthetic generator: .npz with start, end, annot (int64, "
                         "start-sorted), num_docs, length (tools/realistic_index.py writes them); the window is [0, length)")
    ap.add_argument("--membership", action="store_true", help="with --rows-file: membership query (the file holds a membership index)")
    ap.add_argument("--launch", action="store_true",
                    help="start the ranks as a child torch.distributed.run even when --gpus is 1 (N > 1 without "
                         "WORLD_SIZE in the environment does so by itself)")
    ap.add_argument("--headline-only", action="store_true",
                    help="N = 1: skip the legs that time the other resident row formats (rocprofv3 passes: every launch of the "
                         "headline kernel in the trace is then the headline workload)")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000,
                    help="positions of the window the 1-core CPU baseline is timed on (0 = skip)")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process may use, counted WITHOUT initialising one (torch.cuda.device_count() reads the
    topology only on this image; no HIP call, no memo_device_count).  MEMO_BENCH_ASSUME_DEVICES overrides
    it (the CPU test of the launcher)."""
    if os.environ.get("MEMO_BENCH_ASSUME_DEVICES"):
        return int(os.environ["MEMO_BENCH_ASSUME_DEVICES"])
    import torch
    return torch.cuda.device_count()


def launch_ranks(args, argv):
    """`python bench.py --gpus N` as the driver spells it, with no WORLD_SIZE in the environment: this process
    becomes the PARENT of one `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py <the same arguments>` -- a child process (subprocess; never os.exec*, and
    nothing here has touched a GPU) -- relays rank 0's one JSON line on stdout and exits with the child's code.
    Returns None when this process is itself a rank (or a plain N = 1 run)."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None                                         # already a rank of somebody's launcher
    if args.gpus <= 1 and not args.launch:
        return None
    have = visible_gpus()
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but {have} GPU(s) visible on this node\n")
        return 2
    import socket
    import subprocess
    with socket.socket() as sk:                             # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child_argv = [a for a in argv if a != "--launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child_argv
    if os.environ.get("MEMO_BENCH_CHILD_CMD"):              # the CPU test of this function: a stub child
        cmd = json.loads(os.environ["MEMO_BENCH_CHILD_CMD"]) + child_argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL needs it on this pool)
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)          # stderr passes through
    out, _ = proc.communicate()
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{") and ln.rstrip().endswith("}")]
    for ln in out.decode(errors="replace").splitlines():   # anything else a rank printed on fd 1: to stderr
        if ln not in lines:
            sys.stderr.write(ln + "\n")
    if proc.returncode == 0 and len(lines) != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, got {len(lines)}\n")
        return 3
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return proc.returncode


def cpu_baseline(args, num_docs, L, k, membership, gpu_result_slice):
    """The oracle's literal transcription of memo_init + memo_query + argmax, one core,
    on the first `cpu_sample` positions of rank 0's window.  Also serves as the parity check
    of the benchmarked result (same positions)."""
    import numpy as np
    from memo_amd import synth
    from oracle import memo_oracle as oracle        # checker / baseline leg only
    S = min(args.cpu_sample, L)
    if membership:
        S = min(S, 3_000_000)
    if args.rows_file:
        z = np.load(args.rows_file)
        r0, r1 = 0, int(np.searchsorted(z["start"], S + k, side="left"))
        s, e, o = (np.ascontiguousarray(z[c][:r1]) for c in ("start", "end", "annot"))
    else:
        num, den = synth.rows_per_position(num_docs)
        r0, r1 = synth.shard_rows(0, S, k, num, den, L)
        s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, num_docs)
    fn = oracle.membership if membership else oracle.conservation
    t0 = time.perf_counter()
    want = fn(s, e, o, 0, S, k, num_docs, literal=True)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(want, gpu_result_slice(S)))
    base = {"value": S / dt, "unit": "query-positions/s", "cores": 1, "kind": "port",
            "sample": f"first {S} positions of the window ({r1 - r0} rows), literal bool-matrix "
                      f"port of memo_query.py:45-63,70 (oracle_literal_*), {dt:.1f} s",
            "parity_with_gpu_on_sample": ok}
    # the strong CPU baseline of SURVEY.md 8(d)(ii): closed form (no L x N matrix), the window cut
    # into one sub-window per thread (the same sharding rule the GPUs use), all in C (GIL released)
    import concurrent.futures as cf
    threads = max(1, min(os.cpu_count() or 1, 64))
    cuts = np.linspace(0, S, threads + 1).astype(np.int64)
    fn2 = oracle.membership if membership else oracle.conservation

    def part(i):
        a, b = int(cuts[i]), int(cuts[i + 1])
        i0 = int(np.searchsorted(s, a, side="right"))
        i1 = int(np.searchsorted(s, b + k, side="left"))
        return fn2(s[i0:i1], e[i0:i1], o[i0:i1], a, b, k, num_docs, literal=False)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(part, range(threads)))
    dt2 = time.perf_counter() - t0
    ok2 = bool(np.array_equal(np.concatenate(parts), want))
    base["all_cores"] = {"value": S / dt2, "unit": "query-positions/s", "cores": threads, "kind": "port",
                         "sample": f"same {S} positions, closed-form restatement (oracle_closed_*) on {threads} "
                                   f"threads, one sub-window each, {dt2:.2f} s", "equal_to_literal_port": ok2}
    return base


def main():
    args = parse()
    rc = launch_ranks(args, sys.argv[1:])
    if rc is not None:
        raise SystemExit(rc)
    global np, torch, dist, memo_amd, shard, synth
    import numpy as np
    import torch
    import torch.distributed as dist
    import memo_amd
    from memo_amd import shard, synth
    # stdout carries exactly one JSON line.  RCCL and the HIP runtime sometimes print banners
    # on fd 1, so fd 1 is pointed at stderr for the run and the line goes to the saved fd.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus disagree")
    # TEST transport (tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu): RCCL refuses two ranks on one device and this
    # pool gives one GPU per box, so the N > 1 control flow -- link probe, coding choice, root weight, per-step send / receive,
    # decode on rank 0, parity of the gathered result -- can run here only with every rank on GPU 0 (MEMO_BENCH_ONE_DEVICE=1)
    # and gloo carrying the bytes through host memory (MEMO_BENCH_BACKEND=gloo).  The line says so ("test_transport"); its
    # numbers are not bench numbers.
    backend = os.environ.get("MEMO_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("MEMO_BENCH_BACKEND: nccl (RCCL; the bench) or gloo (test transport through host memory)")
    if os.environ.get("MEMO_BENCH_ONE_DEVICE") == "1":
        local = 0
    staged = backend == "gloo"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.force_dist
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            dist.init_process_group(backend, rank=0, world_size=1, **({} if staged else {"device_id": dev}))
        else:
            dist.init_process_group(backend, **({} if staged else {"device_id": dev}))

    def all_reduce_max(x):                  # (device tensor; through host memory on the test transport)
        if not staged:
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            return
        h = x.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MAX)
        x.copy_(h)

    def broadcast0(x):
        if not staged:
            dist.broadcast(x, src=0)
            return
        h = x.cpu()
        dist.broadcast(h, src=0)
        x.copy_(h)

    def gather0(x, bufs):
        if not staged:
            dist.gather(x, bufs, dst=0)
            return
        h = x.cpu()
        hb = [torch.empty_like(h) for _ in range(world)] if rank == 0 else None
        dist.gather(h, hb, dst=0)
        if rank == 0:
            for d_, s_ in zip(bufs, hb):
                d_.copy_(s_)

    num_docs, L, membership = WORKLOADS[args.workload]
    host_rows = None
    if args.rows_file:                       # an index built elsewhere (sequences -> matching statistics -> dap_to_bed)
        if world != 1:
            raise SystemExit("--rows-file is a single-GPU workload")
        z = np.load(args.rows_file)
        host_rows = tuple(np.ascontiguousarray(z[c], dtype=np.int64) for c in ("start", "end", "annot"))
        num_docs, L, membership = int(z["num_docs"]), int(z["length"]), bool(args.membership)
        args.workload = "rows-file"
    k = args.k
    pivot = L * world
    qs, qe = rank * L, (rank + 1) * L
    from memo_amd import _lib
    if args.calibrate:
        _lib.use_ab(True)                   # the PMC calibration kernel lives in libmemo_amd_ab.so only
    # the dense rows answer the unclipped conservation sweep only: k <= 64, num_docs <= 511, >= 1 row per position
    # (membership reads them too when an index holds nothing else, 4 % slower than the 4-byte rows: not a bench format)
    # (256 .. 511 genomes -- BASELINE config 5 -- since round 4: the ninth annot bit in the group's spare byte, uint16 results)
    can_dense = (not membership) and k - 1 <= 63 and ((num_docs <= 255 and not args.wide) or 255 < num_docs <= 511) and num_docs >= 20
    if host_rows is not None:                # the library's own rule, on the rows the window sees
        from memo_amd.index import dense_rows_can_answer
        hs, _, ho = host_rows
        can_dense = not args.wide and len(hs) > 0 and dense_rows_can_answer(len(hs), int(hs[0]), int(hs[-1]), int(ho.max()), k,
                                                                             num_docs, membership)
    if args.rows == "auto":     # the fastest format that can answer: dense rows (back to back they are 13 % ahead of the
        args.rows = "wide" if k - 1 > 255 else ("dense" if can_dense else "packed")     # 4-byte rows: DESIGN.md section 7)
    if args.rows == "dense" and not can_dense:
        raise SystemExit("--rows dense answers conservation with k <= 64 and num_docs <= 511 only")
    if k - 1 > 255:
        args.rows = "wide"                  # packed rows answer k <= 256 only
    # Resident indexes of the same rows, one per row format: the int64 columns as uploaded (24 B/row), the
    # packed query format (memo_index_pack: 4 / 6 B/row, int64 columns dropped) and the dense rows
    # (memo_index_pack_dense, everything else dropped).  SURVEY.md 8(d): the passes that narrow the rows are
    # timed apart from the query -- on the device (HIP event pair around the annot census and the packing
    # kernel inside memo_index_pack, buffers allocated by the first call and reused by the second; an event
    # pair on the same stream around memo_index_pack_dense).
    indexes, pack_pass, dense_pass, packed_fmt = {}, None, None, 0
    others = (world == 1 and not args.force_dist and not args.headline_only)
    formats = [args.rows] + ([f for f in ("wide", "packed", "dense") if f != args.rows and
                              (f != "dense" or can_dense) and (f == "wide" or k - 1 <= 255)] if others else [])
    if args.calibrate and "wide" not in formats:
        formats.append("wide")              # (the PMC calibration kernel streams the int64 columns: 16 known bytes per row)
    one_device = os.environ.get("MEMO_BENCH_ONE_DEVICE") == "1" and world > 1
    for turn in range(world if one_device else 1):       # (test transport: the ranks share one GPU and build one after the other)
      if one_device:
          dist.barrier()
      if one_device and turn != rank:
          continue
      for f in formats:
          if host_rows is not None:
              ixf = memo_amd.DeviceIndex.from_host(*host_rows, device=local)
              r0, r1 = int(np.searchsorted(host_rows[0], qs, side="right")), int(np.searchsorted(host_rows[0], qe + k, side="left"))
          else:
              ixf, (r0, r1) = synth.device_index(qs, qe, k, num_docs, pivot, device=local)
          nrows = r1 - r0
          if f != "wide":
              ixf.pack(keep_wide=True)            # allocates the packed rows
              ixf.pack(keep_wide=True)            # the timed pass: same buffers
              info = ixf.info()
              packed_fmt = info["packed_format"]
              pk_bytes = 6 if packed_fmt == 6 else 4
              pack_bytes = (24 + pk_bytes) * nrows + 2 * pk_bytes * nrows        # + the order inside the buckets: the words read and written again
              pack_pass = {"what": "memo_index_pack: int64 columns -> packed rows, once per index, not part of a query "
                                   "(the layout guessed from a sample of the annot column; pack_rows_kernel reads 24 B, writes "
                                   "%d B per row (format %d) and takes the exact annot census on the way; then the rows' order inside "
                                   "their buckets, in place: interleave_small_kernel)" % (pk_bytes, packed_fmt),
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
              dense_pass = {"what": "memo_index_pack_dense: 4-byte rows -> dense rows, five per 16 bytes (reads 4 B, writes "
                                    "3.2 B per row; includes the hipMalloc of the dense rows), once per index",
                            "ms": dms, "rows": nrows, "bytes": 7.2 * nrows, "GBs": 7.2 * nrows / (dms * 1e-3) / 1e9}
          indexes[f] = ixf
    rows = r1 - r0
    rows_read = {f: rows for f in indexes}
    ix = indexes[args.rows]
    fmt_bytes = {"wide": 24, "packed": 6 if packed_fmt == 6 else 4, "dense": 3.2}
    row_bytes = fmt_bytes[args.rows]
    view_bytes = {}                          # per format: bytes per row of the k-class VIEW the sweep reads, where they differ
    W = (num_docs + 31) // 32
    # result element: membership = W uint32 words; conservation = uint8 when num_docs <= 255 (the
    # same at every N; it also halves what the slices put on xGMI), else uint16
    narrow = (not membership) and num_docs <= 255 and not args.wide
    if membership:
        shape, dtype, b_out = (L, W), torch.int32, 4 * W
    elif narrow:
        shape, dtype, b_out = (L,), torch.uint8, 1
    else:
        shape, dtype, b_out = (L,), torch.int16, 2          # uint16 payload
    # two result buffers: the gather of step i (RCCL stream) overlaps the sweep of step i+1
    nbuf = 2 if multi else 1
    outs = [torch.empty(shape, dtype=dtype, device=dev) for _ in range(nbuf)]
    stream = torch.cuda.current_stream()
    lib = _lib.lib()

    qe_mine = qe                             # (N > 1: rank 0 may sweep less than its whole window, see below)

    def launch(out, ix=ix):
        if membership:
            ix.membership_dev(qs, qe_mine, k, num_docs, out, stream.cuda_stream)
        elif narrow:
            ix.conservation_u8_dev(qs, qe_mine, k, num_docs, out, stream.cuda_stream)
        else:
            ix.conservation_dev(qs, qe_mine, k, num_docs, out, stream.cuda_stream)

    # One launch per resident format before anything is timed: the first conservation query of a k class on the dense rows
    # builds their k-class VIEW (the rows whose overlap is below the class's cap -- 2, 4, 6 ... 32 --: all that can write at this k; memo_query.py:49
    # drops the others per query, the library once per index and class -- memo_index_info_t.last_rows_read) and the tile
    # table; both stay with the index.  What each format's sweep reads is what its roofline is priced on.
    view_pass = place_pass = None
    unplaced = None                          # the headline sweep on the view BEFORE its rows were placed (what a query-built view is at first)
    full_rows = {}
    prepared = {}
    for f, ixf in indexes.items():
        if f != "wide":
            # memo_index_prepare: the k-class view and the tile table NOW (a host that sweeps one k over many windows calls it
            # once; without it the view is built by the query that finds it has become worth its pass: MEMO_OPT_BUILD_COST_PCT)
            # -- timed on the device by the library.  Dense rows: in the two steps a sequence of queries would take -- first the
            # view with its rows in the order they come (MEMO_OPT_VIEW_PLACES 0), then again with the rows' places inside their
            # groups chosen against LDS bank conflicts -- each timed, and the sweep on the first timed too.
            if f == "dense" and f == args.rows and not membership:
                ixf.set_option(5, 0)
            prepared[f] = ixf.prepare(k, num_docs, membership)
            inf = ixf.info()
            if f == args.rows and inf["last_view_ms"] > 0:
                view_pass = {"what": "k-class view of the rows the sweep reads (rows whose overlap is below the class's cap: all that can "
                                     "write at this k), built by memo_index_prepare (else: by the query that finds the class's queries "
                                     "have lost more to its absence than it costs) -- count, scan, one fused pass over "
                                     "the rows -- once per index and class, kept",
                             "ms": float(inf["last_view_ms"]), "rows_in": int(inf["dense_row_count"] if f == "dense" else inf["rows"]),
                             "rows_kept": int(inf["last_rows_read"]), "device_bytes_taken_by_prepare": int(prepared[f])}
                # ... and once more on memory the process has held before, like row_format_pass's second call: the pass is timed with
                # its allocations, and a first hipMalloc of a fresh 0.85 GB region has taken the driver 350 ms on one box of the pool
                # (gpurun r5valid) where the pass takes 1.7
                ixf.set_option(1, 0)
                ixf.set_option(1, 1)
                ixf.check()
                ixf.prepare(k, num_docs, membership)
                again = float(ixf.info()["last_view_ms"])
                if again > 0:
                    view_pass.update({"ms_first_build": view_pass["ms"], "ms_second_build": again, "ms": min(view_pass["ms"], again),
                                      "timed_with": "HIP event pair around the whole pass inside memo_index_prepare (its allocations and two host "
                                                    "waits included); built twice, the views dropped in between: `ms` is the smaller"})
            if f == "dense" and f == args.rows and not membership:
                ixf.set_option(5, 1)
                if view_pass:
                    if world == 1 and not args.force_dist and not args.headline_only:
                        # the sweep on the view as a sequence of queries first gets it (300 launches, the last 100 timed)
                        ev_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(300)]
                        for a_, b_ in ev_:
                            a_.record(stream)
                            launch(outs[0], ixf)
                            b_.record(stream)
                        torch.cuda.synchronize()
                        ms_u = sorted(a_.elapsed_time(b_) for a_, b_ in ev_[200:])
                        unplaced = {"kernel_ms_median": float(ms_u[len(ms_u) // 2]), "rows_read": int(ixf.info()["last_rows_read"])}
                    ixf.prepare(k, num_docs, membership)
                    inf = ixf.info()
                    if inf["last_view_ms"] > 0:
                        place_pass = {"what": "the same view built again with the place of every row inside its 16-byte group chosen against LDS "
                                              "bank conflicts (memo_view.hip: view_place_bucket) -- by memo_index_prepare at once, by queries "
                                              "once the class's queries have lost to the plain view what this pass costs (MEMO_OPT_VIEW_PLACES)",
                                      "ms": float(inf["last_view_ms"])}
        launch(outs[0], ixf)
        torch.cuda.synchronize()
        inf = ixf.info()
        full_rows[f] = int(inf["dense_row_count"]) if f == "dense" else rows
        if inf["last_rows_read"]:
            rows_read[f] = int(inf["last_rows_read"])
        if f == "dense" and inf.get("last_view_rows_per_group") == 6:
            view_bytes[f] = 16.0 / 6.0               # a view of six rows per group: bytes per row (slot) the sweep reads
            if f == args.rows:
                row_bytes = view_bytes[f]
    # What travels to rank 0.  A slice's own xGMI link is what bounds N > 1, so uint8
    # conservation slices go in a lossless transport coding: "runs" (one bit per position + a byte per change of
    # value, memo_transport_runs_*), "dense" (2 bits per position + a nibble per value outside 1..3,
    # memo_transport_dense_*), "nibble" (one nibble per position, memo_transport_*), or plain bytes.  Every rank sizes the codings on its own first result (the same query runs every
    # step); a coding is usable when every rank's slice fits it.  Fewer bytes is not all: rank 0 decodes
    # world - 1 slices per step, so among the usable codings it takes the one whose modelled step
    #     max(sweep + encode [a peer], sweep + (world - 1) * decode of one slice [rank 0],
    #         wire bytes / 75 GB/s [a peer's link])           (memo_amd.shard.pick_coding)
    # is shortest, with sweep, encode and decode timed here on rank 0.
    coding, b_cap = "plain", 0
    cap = max(L // 256, 1024)
    choice = None
    # N > 1: who is here, and what a peer's link into rank 0 delivers -- measured, not assumed: a gather of one
    # plain slice per rank (what --plain-gather would send every step), 3 repetitions after one untimed.  Every
    # peer sends over its own xGMI link at once, so slice bytes / gather time is the per-peer rate the step model
    # needs (shard.XGMI_LINK_BYTES_PER_S is only the stand-in for a run with a single rank, where nothing travels).
    link_measured, ranks_seen, link_probe = None, None, None
    if multi:
        props = torch.cuda.get_device_properties(local)
        me = {"rank": rank, "device": local, "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None),
              "uuid": str(getattr(props, "uuid", ""))}
        seen = [None] * world
        dist.all_gather_object(seen, me)
        ranks_seen = {"world_size": dist.get_world_size(), "ranks": seen,
                      "distinct_devices": len({(r["device"], r["pci_bus_id"], r["uuid"]) for r in seen})}
        launch(outs[0])
        probe = outs[0].view(torch.uint8).reshape(-1)
        probe = probe[:min(probe.numel(), 1 << 28)]
        bufs = [torch.empty_like(probe) for _ in range(world)] if rank == 0 else None
        gather0(probe, bufs)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t_g0 = time.perf_counter()
        for _ in range(3):
            gather0(probe, bufs)
        torch.cuda.synchronize()
        t_g = torch.tensor([(time.perf_counter() - t_g0) / 3], dtype=torch.float64, device=dev)
        all_reduce_max(t_g)
        link_probe = {"bytes_per_rank": probe.numel(), "gather_ms": float(t_g[0]) * 1e3, "reps": 3}
        if world > 1:
            link_measured = probe.numel() / float(t_g[0])
        del bufs, probe
    link = link_measured or shard.XGMI_LINK_BYTES_PER_S
    root_weight = 1.0 if args.root_weight == "auto" else min(max(float(args.root_weight), 0.01), 1.0)
    vbytes = 1 if narrow else 2                  # bytes per conservation value
    codable = multi and not membership and not args.plain_gather
    runs_pack_fn = lib.memo_transport_runs_pack_dev if narrow else lib.memo_transport_runs16_pack_dev
    runs_unpack_fn = lib.memo_transport_runs_unpack_dev if narrow else lib.memo_transport_runs16_unpack_dev
    if codable:
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

        t_sweep = timed(lambda: launch(outs[0]))
        scratch = torch.empty(L * vbytes, dtype=torch.uint8, device=dev)
        usable = {"plain": (L * vbytes, 0.0, 0.0)}                         # coding -> (wire bytes, decode s, encode s)
        if narrow:                               # (the dense and nibble codings carry uint8 values only)
            # dense
            trial_b = ((L // 2 + 4 * (L // 32768 + 1)) + 3) & ~3              # every position an escape
            probe = torch.empty(lib.memo_transport_dense_bytes(L, trial_b, cap), dtype=torch.uint8, device=dev)
            _lib.check(lib.memo_transport_dense_pack_dev(outs[0].data_ptr(), L, trial_b, cap, probe.data_ptr(), local,
                                                         stream.cuda_stream))
            _lib.check(lib.memo_transport_dense_stats(probe.data_ptr(), local, stream.cuda_stream, C.byref(found),
                                                      C.byref(have), C.byref(taken), C.byref(room)))
            need = torch.tensor([taken.value, found.value], dtype=torch.int64, device=dev)
            all_reduce_max(need)
            b_cap = (int(need[0].item()) + 4096 + 3) & ~3                      # exact + slack
            dense_cap = int(need[1].item()) + 1024
            if not args.nibble_gather:
                t = timed(lambda: _lib.check(lib.memo_transport_dense_unpack_dev(
                    probe.data_ptr(), L, trial_b, cap, scratch.data_ptr(), local, stream.cuda_stream)))
                tp = timed(lambda: _lib.check(lib.memo_transport_dense_pack_dev(
                    outs[0].data_ptr(), L, trial_b, cap, probe.data_ptr(), local, stream.cuda_stream)))
                usable["dense"] = (lib.memo_transport_dense_bytes(L, b_cap, dense_cap), t, tp)
            # nibble
            probe = torch.empty(lib.memo_transport_bytes(L, cap), dtype=torch.uint8, device=dev)
            _lib.check(lib.memo_transport_pack_dev(outs[0].data_ptr(), L, cap, probe.data_ptr(), local, stream.cuda_stream))
            _lib.check(lib.memo_transport_exceptions(probe.data_ptr(), local, stream.cuda_stream, C.byref(found), C.byref(have)))
            need = torch.tensor([found.value], dtype=torch.int64, device=dev)
            all_reduce_max(need)
            nibble_cap = int(need.item()) + int(need.item()) // 16 + 1024          # what the ranks found + slack
            if nibble_cap * 8 <= L // 8:                                       # else the list outweighs the saving
                t = timed(lambda: _lib.check(lib.memo_transport_unpack_dev(
                    probe.data_ptr(), L, scratch.data_ptr(), local, stream.cuda_stream)))
                tp = timed(lambda: _lib.check(lib.memo_transport_pack_dev(
                    outs[0].data_ptr(), L, cap, probe.data_ptr(), local, stream.cuda_stream)))
                usable["nibble"] = (lib.memo_transport_bytes(L, nibble_cap), t, tp)
        # runs: one bit per position + one value (one or two bytes: memo_transport_runs16_*) per change of value -- the
        # coding that carries BASELINE config 5's uint16 slices too (67 MB plain against ~9 MB at k = 31)
        trial_r = (L * vbytes + 4 * (L // 32768 + 1) + 3) & ~3            # every position a change
        probe = torch.empty(lib.memo_transport_runs_bytes(L, trial_r), dtype=torch.uint8, device=dev)
        _lib.check(runs_pack_fn(outs[0].data_ptr(), L, trial_r, probe.data_ptr(), local, stream.cuda_stream))
        _lib.check(lib.memo_transport_runs_stats(probe.data_ptr(), local, stream.cuda_stream, C.byref(taken), C.byref(room)))
        need = torch.tensor([taken.value], dtype=torch.int64, device=dev)
        all_reduce_max(need)
        runs_cap = (int(need.item()) + int(need.item()) // 16 + 4096 + 3) & ~3       # what the ranks needed + slack
        if not args.nibble_gather:
            t = timed(lambda: _lib.check(runs_unpack_fn(
                probe.data_ptr(), L, trial_r, scratch.data_ptr(), local, stream.cuda_stream)))
            tp = timed(lambda: _lib.check(runs_pack_fn(
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
        best, model = shard.pick_coding(world, t_sweep, usable, link=link)
        w_best = 1.0
        if args.root_weight == "auto":
            best, w_best, _ = shard.pick_plan(world, t_sweep, usable, link=link)
        else:
            w_best = min(max(float(args.root_weight), 0.01), 1.0)
        if args.coding in usable:
            best = args.coding
        names = sorted(usable)
        pick = torch.tensor([names.index(best), int(round(w_best * 1000))], device=dev)
        broadcast0(pick)                                        # rank 0's timings decide for everybody
        coding = names[int(pick[0].item())]
        root_weight = int(pick[1].item()) / 1000.0
        if coding == "dense":
            cap = dense_cap
        elif coding == "nibble":
            cap = nibble_cap
        elif coding == "runs":
            b_cap = runs_cap
        choice = {"picked": coding, "root_weight": root_weight, "sweep_ms": t_sweep * 1e3,
                  "link_GBs_used": link / 1e9,
                  "link_source": "measured in this run (plain gather of one slice per rank)" if link_measured else
                                 "assumed (one rank: nothing travels)",
                  "modelled_step_ms": shard.modelled_step(world, t_sweep, *usable[coding], link=link,
                                                          root_weight=root_weight) * 1e3,
                  "candidates": {c: {"wire_bytes": int(usable[c][0]), "decode_ms_per_slice": usable[c][1] * 1e3, "encode_ms": usable[c][2] * 1e3,
                                     "decode_ms_all_slices_one_launch": usable[c][3] * 1e3 if len(usable[c]) > 3 else None,
                                     "modelled_step_ms": model[c] * 1e3} for c in names}}
    nibble = coding != "plain"              # (name kept: "the slices travel coded")
    # rank 0 sweeps the first root_weight of its window (a multiple of 8 positions); everybody else all of it
    L_mine = L if rank != 0 else max(8, int(L * root_weight) // 8 * 8)
    for o in outs:                          # (every result buffer holds a whole-window result behind L_mine)
        launch(o)
    qe_mine = qs + L_mine
    positions_per_step = L * (world - 1) + (max(8, int(L * root_weight) // 8 * 8) if multi else L)

    def pack(src, wire):
        if coding == "runs":
            _lib.check(runs_pack_fn(src.data_ptr(), L, b_cap, wire.data_ptr(), local, stream.cuda_stream))
        elif coding == "dense":
            _lib.check(lib.memo_transport_dense_pack_dev(src.data_ptr(), L, b_cap, cap, wire.data_ptr(), local,
                                                         stream.cuda_stream))
        else:
            _lib.check(lib.memo_transport_pack_dev(src.data_ptr(), L, cap, wire.data_ptr(), local, stream.cuda_stream))

    def unpack(wire, dst):
        if coding == "runs":
            _lib.check(runs_unpack_fn(wire.data_ptr(), L, b_cap, dst.data_ptr(), local, stream.cuda_stream))
        elif coding == "dense":
            _lib.check(lib.memo_transport_dense_unpack_dev(wire.data_ptr(), L, b_cap, cap, dst.data_ptr(), local,
                                                           stream.cuda_stream))
        else:
            _lib.check(lib.memo_transport_unpack_dev(wire.data_ptr(), L, dst.data_ptr(), local, stream.cuda_stream))

    if coding == "runs":
        wires = [torch.empty(lib.memo_transport_runs_bytes(L, b_cap), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    elif coding == "dense":
        wires = [torch.empty(lib.memo_transport_dense_bytes(L, b_cap, cap), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    elif coding == "nibble":
        wires = [torch.empty(lib.memo_transport_bytes(L, cap), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    else:
        wires = [o.view(torch.uint8).reshape(-1) for o in outs]  # RCCL has no 16-bit integer type
    roots = [[torch.empty_like(wires[0]) for _ in range(world)] if (multi and rank == 0) else None
             for _ in range(nbuf)]
    if multi and rank == 0 and coding == "plain":
        for b in range(nbuf):               # plain bytes: rank 0 sweeps straight into its place of the gathered result
            roots[b][0] = wires[b]
    # rank 0: the gathered slices in result form (decoded when they travelled as nibbles)
    decoded = [[torch.empty(L, dtype=dtype, device=dev) for _ in range(world)] if (nibble and rank == 0) else None
               for _ in range(nbuf)]
    pending = [None] * nbuf
    host_side = [None] * nbuf               # (test transport only)
    # rank 0's own slice never travels: its sweep writes straight into the gathered result (no pack, no
    # unpack; the wire it contributes to the gather is ignored).  --code-own-slice codes it like a
    # peer's, so that one GPU can exercise the whole path.
    skip_own = nibble and rank == 0 and not args.code_own_slice
    if skip_own:
        outs = [decoded[b][0] for b in range(nbuf)]

    def send_slices(b):
        """every peer's wire b -> rank 0 (roots[b][g]): grouped RCCL send / recv, one per peer link; rank 0's own slice
        never travels (dist.gather would copy it through RCCL's copy kernel, next to the sweep: 10 % of a step at
        N = 1, gpurun r3a)."""
        if rank == 0:
            if nibble and args.code_own_slice:
                roots[b][0].copy_(wires[b])     # (validation mode: rank 0's slice coded like a peer's)
            if staged:
                host_side[b] = [None] + [torch.empty(wires[b].numel(), dtype=torch.uint8) for _ in range(1, world)]
            ops = [dist.P2POp(dist.irecv, host_side[b][g] if staged else roots[b][g], g) for g in range(1, world)]
        else:
            if staged:
                host_side[b] = wires[b].cpu()   # (test transport: the wire leaves through host memory)
            ops = [dist.P2POp(dist.isend, host_side[b] if staged else wires[b], 0)]
        return dist.batch_isend_irecv(ops) if ops else []

    def finish(b):                          # gather b done -> (root) slices back in result form
        if pending[b] is None:
            return
        for work in pending[b]:
            work.wait()
        pending[b] = None
        if staged and rank == 0:
            for g in range(1, world):
                roots[b][g].copy_(host_side[b][g])
        if nibble and rank == 0:
            first = 1 if skip_own else 0
            if coding == "runs" and world - first > 0:      # every slice of the step by one launch
                cnt = world - first
                ws = (C.c_void_p * cnt)(*[roots[b][g].data_ptr() for g in range(first, world)])
                os_ = (C.c_void_p * cnt)(*[decoded[b][g].data_ptr() for g in range(first, world)])
                _lib.check(lib.memo_transport_runs_unpack_many_dev(ws, os_, cnt, L, b_cap, vbytes, local, stream.cuda_stream))
            else:
                for g in range(first, world):
                    unpack(roots[b][g], decoded[b][g])

    def step(i):
        b = i % nbuf
        finish(b)                           # buffer b is free once its previous gather is done
        launch(outs[b])
        if multi:                           # result slices -> rank 0 over xGMI (RCCL send/recv)
            if nibble and not skip_own:
                pack(outs[b], wires[b])
            pending[b] = send_slices(b)

    def drain():
        for b in range(nbuf):
            finish(b)
        torch.cuda.synchronize()

    def kernel_name(which):
        if which == "dense" and not membership:
            inf = indexes[which].info()
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
            return ("sweep_membership_planes_kernel<" if num_docs <= 512 else "sweep_membership_runs_kernel<") + rows_t + ", ...>"
        # the library says which kernel family answered the last conservation query on this index (it picks the
        # level arrays per query from k and the overlap lengths of the rows: memo_sweep_cons.hip, pick_levels)
        family = {1: "sweep_conservation_kernel<", 2: "sweep_conservation_halo_kernel<", 3: "sweep_conservation_r4_kernel<",
                  4: "sweep_conservation_mixed_kernel<"}[indexes[which].info()["last_sweep"]]
        return family + rows_t + ", ...>"

    def per_step(fn, n):
        """n launches with a HIP event pair each -> ms per launch (list)"""
        ev_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a_, b_ in ev_:
            a_.record(stream)
            fn()
            b_.record(stream)
        torch.cuda.synchronize()
        return [a_.elapsed_time(b_) for a_, b_ in ev_]

    if args.calibrate:
        indexes["wide"].debug_stream_rows(stream.cuda_stream)

    # (1) the same query on the OTHER row format, for the record -- and first, so that the headline's
    # timed region does not sit in the clock ramp of a cold device (the driver runs --steps 20 --warmup 5)
    def settle(fn):
        """untimed launches until the device has settled on this kernel -- a change of kernels sets off a swing of the
        clocks (the sweep runs at the 1400 W power cap: fast for a few launches, then slower than its steady state,
        then back over some thirty launches; profiles/r02_dense_rows_ab.txt): batches of 20 until two in a row are
        within 1 % of the one before them; at least 5 batches, at most 1.5 s of them"""
        prev, calm, n, t_ramp = None, 0, 0, time.perf_counter()
        while True:
            cur = float(np.mean(per_step(fn, 20)))
            n += 20
            calm = calm + 1 if prev is not None and abs(cur - prev) <= 0.01 * prev else 0
            prev = cur
            if (calm >= 2 and n >= 100) or time.perf_counter() - t_ramp > 1.5:
                return {"launches": n, "ms": (time.perf_counter() - t_ramp) * 1e3}

    other = []
    legs = [(w, w, None) for w in formats[1:]]
    if others and args.rows != "wide" and rows_read[args.rows] != full_rows[args.rows]:
        # the same kernel on ALL the rows of the headline format (MEMO_OPT_VIEWS off: no k-class view), for the record: what round 2 timed
        legs.append((args.rows, args.rows + ", all rows (no k-class view)", full_rows[args.rows]))
    if others and args.rows == "dense" and view_bytes.get("dense"):
        # the view of FIVE rows per group (MEMO_OPT_VIEW_ROWS 5: rounds 3-4's headline; 3.2 B per row, no padding), for the record
        legs.append(("dense", "dense, the k-class view as groups of five rows (MEMO_OPT_VIEW_ROWS 5)", -5))
    if others and not membership and narrow:
        legs.append((args.rows, args.rows + ", uint16 results (memo_query_conservation_dev: the ABI's primary result type)", -16))
    for which, label, all_rows in legs:
        ob = fmt_bytes[which] if all_rows and all_rows > 0 else view_bytes.get(which, fmt_bytes[which])
        fn2, out2, b_out2, read2 = (lambda: launch(outs[0], indexes[which])), None, b_out, None
        if all_rows and all_rows > 0:
            indexes[which].set_option(1, 0)         # MEMO_OPT_VIEWS = 0: the views go, every sweep reads all the rows
        elif all_rows == -5:
            indexes[which].set_option(1, 0)         # (one view resident at a time, as in the headline's own leg)
            indexes[which].set_option(1, 1)
            indexes[which].set_option(4, 5)
            indexes[which].prepare(k, num_docs, membership)
            ob = fmt_bytes[which]
        elif all_rows == -16:
            out2 = torch.empty(L, dtype=torch.int16, device=dev)
            b_out2 = 2
            fn2 = lambda: indexes[which].conservation_dev(qs, qe_mine, k, num_docs, out2, stream.cuda_stream)
        try:
            settle(fn2)
            ms2 = per_step(fn2, max(args.steps, 20))
            indexes[which].check(stream.cuda_stream)
            name2 = kernel_name(which)
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
        alg2 = ob * (all_rows or rows_read[which]) + b_out2 * L
        med2 = float(np.median(ms2))
        other.append({"rows": label, "row_bytes": ob, "rows_read": all_rows or rows_read[which], "kernel": name2,
                      "kernel_ms": float(np.mean(ms2)),
                      "kernel_ms_median": med2, "kernel_ms_min": float(np.min(ms2)),
                      "query_positions_per_s": L / (med2 * 1e-3), "algorithmic_bytes": alg2,
                      "achieved_GBs": alg2 / (med2 * 1e-3) / 1e9, "frac": alg2 / (med2 * 1e-3) / 1e9 / HBM_PEAK_GBS})

    # (2) settling on the headline kernel (see settle())
    ramp = settle(lambda: launch(outs[0]))

    for i in range(args.warmup):
        step(i)
    drain()
    ix.check(stream.cuda_stream)           # raises if the kernel flagged a bad row

    # HIP events on the launch stream: ONE pair around the K timed steps at N = 1 (the sweeps run back to back;
    # two event packets between every two launches cost the timed region 5-7 us per step) and none at all inside
    # the timed region at N > 1 (or --force-dist): there a step is sweep + coding + gather and the wall clock
    # between the barriers is what counts; the sweep alone is timed afterwards, K launches with a pair each.
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not multi:
        ev[0].record(stream)
    for i in range(args.steps):
        step(i)
    if not multi:
        ev[1].record(stream)
    drain()                                  # (torch.cuda.synchronize(): this rank's K steps are done, rank 0's incl. every slice received)
    dt = time.perf_counter() - t0            # N > 1: the MAX over ranks below is the barrier-to-barrier time of the job without
    if multi:                                # the closing barrier's own latency (~0.1 ms of RCCL against K = 20 steps of 0.2 ms)
        dist.barrier()
    ix.check(stream.cuda_stream)
    out = outs[(args.steps - 1) % nbuf]
    # per-step spread (and, N > 1, the sweep's own time): K more sweeps with an event pair each, outside the
    # timed region
    per = per_step(lambda: launch(outs[args.steps % nbuf]), args.steps)
    kern_ms = float(np.mean(per)) if multi else ev[0].elapsed_time(ev[1]) / args.steps
    if multi:
        launch(outs[args.steps % nbuf])     # (leave every buffer holding a complete result)
        t = torch.tensor([dt, kern_ms], dtype=torch.float64, device=dev)
        all_reduce_max(t)
        dt, kern_ms = float(t[0]), float(t[1])

    if rank == 0:
        # bytes of the row layout the timed kernel reads + the result it writes
        b_alg = row_bytes * rows_read[args.rows] + b_out * L
        achieved = b_alg / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": "query-positions/sec (chr window, k=%d)" % k,
            "value": positions_per_step * args.steps / dt,
            "unit": "query-positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64" if args.rows == "wide" else "u32",
            "data": "synthetic" if host_rows is None else "synthetic sequences (random pivot + mutated copies), real matching statistics",
            "config": {"workload": (f"BASELINE config {args.workload[1:]}: synthetic {num_docs}-genome index, "
                                    if host_rows is None else
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
                       "row_bytes": row_bytes, "rows_read": rows_read[args.rows],
                       "rows_read_note": None if args.rows == "wide" or rows_read[args.rows] == full_rows.get(args.rows) else
                                         f"the sweep reads the k-class view of the {args.rows} rows: the {rows_read[args.rows]} of "
                                         f"{full_rows.get(args.rows)} rows that can write at k = {k} (memo_query.py:49 drops the others per "
                                         "query; the library once per index and class -- memo_index_prepare, else the query that finds the view worth its pass: dense_view_pass); "
                                         "`roofline` is priced on the rows read, other_row_formats has the same kernel on all the rows",
                       "result_bytes_per_position": b_out,
                       "row_format_choice": "--rows auto = the format that answers this query fastest: the dense rows where they "
                                            "can (conservation, k <= 64, num_docs <= 511), else the 4- / 6-byte rows (k <= 256), "
                                            "else int64.  roofline is priced on the bytes of the format read (fewer bytes per row "
                                            "lower `frac` at the same speed); the other resident formats are timed in "
                                            "other_row_formats",
                       "row_format_pass": pack_pass, "dense_format_pass": dense_pass, "dense_view_pass": view_pass,
                       "dense_view_place_pass": place_pass,
                       "clock_ramp": {"what": "untimed headline launches before the warm-up steps, in batches of 20 until two "
                                              "batches in a row are within 1 % of the one before them", **ramp},
                       "sharding": f"window split into {world} contiguous sub-windows, slices gathered to rank 0 "
                                   f"over RCCL (double-buffered: gather i overlaps sweep i+1); rank 0 sweeps "
                                   f"{root_weight:g} of a share ({positions_per_step} positions per step in all)"
                                   if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "kernel": kernel_name(args.rows),
                         "kernel_ms": kern_ms, "kernel_ms_median": float(np.median(per)),
                         "kernel_ms_min": float(np.min(per)), "algorithmic_bytes": b_alg,
                         "timed_with": "HIP events on the launch stream: one pair around the K timed steps (kernel_ms "
                                       "= mean); median / min from K further steps with a pair each" if not multi else
                                       "HIP events on the launch stream: K sweeps with a pair each, after the timed "
                                       "region (kernel_ms = mean, slowest rank)"},
        }
        prof = os.path.join(ROOT, "profiles", "traffic.json")    # PMC passes are separate runs
        if os.path.exists(prof):
            key = f"{args.workload}_{args.rows}" if host_rows is None else \
                f"{os.path.basename(os.path.dirname(os.path.abspath(args.rows_file)))}_{'memb' if membership else 'cons'}_k{k}_{args.rows}"
            allt = json.load(open(prof))
            tj = allt.get(f"{key}_k{k}") or allt.get(key)            # (entries of a k other than the workload's default carry it in their key)
            if tj and tj.get("result_bytes_per_position") == b_out and tj.get("algorithmic_bytes") is not None and \
                    abs(tj["algorithmic_bytes"] - b_alg) <= 0.01 * b_alg:   # same kernel instantiation, same rows read (an entry that
                                                                            # does not say what it measured matches nothing)
                res["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
                res["roofline"]["traffic_source"] = (
                    "profiles/traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                    "(%s), not measured in this run" % tj.get("source", "tools/pmc_summary.py"))
            elif tj:                                                    # (a stale entry must not vanish silently)
                sys.stderr.write(f"bench.py: profiles/traffic.json[{key}] is for {tj.get('algorithmic_bytes')} algorithmic bytes / "
                                 f"{tj.get('result_bytes_per_position')} B results, this run has {b_alg} / {b_out}: roofline.traffic left empty\n")
        if multi:
            # what the same run delivers when the result slices stay on their GPUs (no root):
            # every rank's sweep time from its own HIP events, slowest rank counts
            res["without_gather"] = {"value": positions_per_step / (kern_ms * 1e-3), "unit": "query-positions/s",
                                     "note": "aggregate of the per-rank sweeps alone (max kernel_ms over ranks); "
                                             "`value` above includes delivering every slice to rank 0, which is "
                                             f"bound by {wires[0].numel() / 1e6:.0f} MB per peer link per step"}
        if other:
            res["other_row_formats"] = other
            allr = [o for o in other if "all rows" in o["rows"]]
            if allr and view_pass:     # what the same resident index delivers BEFORE its k-class view exists, next to `value`
                first_ms = unplaced["kernel_ms_median"] if unplaced else kern_ms     # the sweep on the view as queries first get it
                gain_ms = allr[0]["kernel_ms_median"] - first_ms
                res["resident_index_without_view"] = {
                    "value": allr[0]["query_positions_per_s"], "unit": "query-positions/s", "kernel_ms_median": allr[0]["kernel_ms_median"],
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
        if world == 1 and args.cpu_sample > 0 and others and "wide" in indexes and host_rows is None and not membership:
            # THE REFERENCE'S SEAM (memo_query.py:103-105: memo_init + memo_query + the argmax of print_res): memo_conservation --
            # three host int64 columns in, a host uint16 result out; what a maintainer who applies INTEGRATION.md section 2 gets.
            # PCIe and the host's packing pass inclusive: never `value`.  The columns come back from the resident int64 index
            # (12 GB for config 3); three calls, the first pays the pinned ring and the worker pool.
            try:
                ds, de, do = indexes["wide"].columns()
                cols = [np.empty(rows, np.int64) for _ in range(3)]
                for h_, d_ in zip(cols, (ds, de, do)):
                    _lib.check(lib.memo_dev_download(local, h_.ctypes.data, d_, h_.nbytes, None))
                ms_seam, seam = [], None
                for _ in range(3):
                    t_s = time.perf_counter()
                    seam = memo_amd.conservation(cols[0], cols[1], cols[2], qs, qe, k, num_docs, device=local)
                    ms_seam.append((time.perf_counter() - t_s) * 1e3)
                got = out.cpu().numpy()
                same = bool(np.array_equal(seam, got.view(np.uint8).astype(np.uint16) if narrow else got.view(np.uint16)))
                from memo_amd.index import dense_rows_can_answer
                dense_in = dense_rows_can_answer(rows, int(cols[0][0]), int(cols[0][-1]), 0, k, num_docs, False)
                res["one_shot_seam"] = {
                    "what": "memo_conservation (include/memo_amd.h): host int64 columns in, host uint16 result out -- the three calls "
                            "memo_query.py:103-105 make; rows packed on the host into pinned memory, PCIe, sweep, result back; the "
                            "index is built and destroyed inside the call",
                    "ms": float(min(ms_seam[1:])), "ms_calls": [float(x) for x in ms_seam], "value": L / (min(ms_seam[1:]) * 1e-3),
                    "unit": "query-positions/s", "rows": rows, "host_bytes_in": 24 * rows, "host_bytes_out": 2 * L,
                    "way_in": ("host packer -> dense rows (3.2 B per row over PCIe)" if dense_in else "host packer -> 4-byte words"),
                    "host_threads": int(os.environ.get("MEMO_HOST_THREADS", min(os.cpu_count() or 1, 32))),
                    "host_cpus_visible": os.cpu_count(), "parity": same}
                del cols, seam
            except Exception as exc:                         # (the seam's line must not cost the run its headline)
                res["one_shot_seam"] = {"error": repr(exc)}
        if world == 1 and args.cpu_sample > 0:
            def gpu_slice(S):
                h = out[:S].cpu().numpy()
                if membership:
                    return h.view(np.uint32)
                return h.view(np.uint8).astype(np.uint16) if narrow else h.view(np.uint16)
            res["cpu_baseline"] = cpu_baseline(args, num_docs, L, k, membership, gpu_slice)
        if multi:
            # the gathered slice of the LAST rank, checked against the oracle on a small sample
            from oracle import memo_oracle as oracle        # checker only
            S, g = 200_000, world - 1
            num, den = synth.rows_per_position(num_docs)
            a = g * L
            sr0, sr1 = synth.shard_rows(a, a + S, k, num, den, pivot)
            s_, e_, o_ = oracle.synth_rows(sr0, sr1 - sr0, num, den, num_docs)
            last = (args.steps - 1) % nbuf
            got = (decoded[last][g] if nibble else roots[last][g].view(dtype).reshape(shape))[:S].cpu().numpy()
            if membership:
                ok = np.array_equal(got.view(np.uint32), oracle.membership(s_, e_, o_, a, a + S, k, num_docs, literal=False))
            else:
                want = oracle.conservation(s_, e_, o_, a, a + S, k, num_docs, literal=False)
                ok = np.array_equal(got.view(np.uint8 if narrow else np.uint16).astype(np.uint16), want)
            res["gather_parity_sample"] = {"rank": g, "positions": S, "equal_to_oracle": bool(ok)}
            if coding != "plain":       # every gathered slice of the last step decodes to the sender's own result?
                whole = True            # (rank 0's own slice against its buffer; the peers' wires report their fill)
                own = decoded[last][0]
                whole &= bool(torch.equal(own, outs[last].reshape(-1)))
                for g2 in range(1 if skip_own else 0, world):
                    head = roots[last][g2][:16].cpu().numpy().view(np.uint32)
                    # head words: runs = (B taken, B capacity, 0, 0); dense = (exceptions, their capacity, B taken, B capacity);
                    # nibble = (exceptions, their capacity, overflow flag, -)
                    whole &= bool(head[0] <= head[1]) and bool(head[2] <= head[3] if coding == "dense" else head[2] == 0)
                res["gather_parity_sample"]["every_slice_complete"] = whole
            res["config"]["gather_payload"] = (
                f"runs coding: 1 bit per position + {b_cap} B for {vbytes} byte(s) per change of value "
                f"({wires[0].numel()} B per slice, {8 * wires[0].numel() / L:.2f} bits per position)"
                if coding == "runs" else
                f"dense coding: 2 bits per position + {b_cap} B of escape nibbles + {cap} exception slots "
                f"({wires[0].numel()} B per slice, {8 * wires[0].numel() / L:.2f} bits per position)"
                if coding == "dense" else
                f"nibble per position + {cap} exception slots ({wires[0].numel()} B per slice)" if coding == "nibble" else
                f"plain result bytes ({wires[0].numel()} B per slice)")
            if choice:
                choice["measured_step_ms"] = dt / args.steps * 1e3
                res["config"]["gather_coding_choice"] = choice
            res["link_GBs_measured"] = link_measured / 1e9 if link_measured else None
            res["link_GBs_assumed_when_unmeasured"] = shard.XGMI_LINK_BYTES_PER_S / 1e9
            res["link_probe"] = link_probe
            res["ranks_seen"] = ranks_seen
            if staged:
                res["test_transport"] = ("gloo through host memory" + (", every rank on GPU 0" if os.environ.get("MEMO_BENCH_ONE_DEVICE") == "1" else "") +
                                         ": a run of the N > 1 control flow, NOT a bench line (value, link and step times mean nothing)")
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    for x in indexes.values():
        x.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
