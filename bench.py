#!/usr/bin/env python3
"""bench.py -- query-positions/sec of the windowed k-mer conservation sweep on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1 either way: `python bench.py --gpus N ...` starts its own ranks -- one child
     `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, whose one JSON line and exit
     code it relays -- or start it under torch.distributed.run yourself; see launch_ranks())

Workload (BASELINE.json config 3 by default): synthetic pangenome, 100 genomes, 5 rows per pivot position (density 0.05
per genome-position), one window of 10^8 positions per GPU, k = 31, conservation.  A "step" is one query of the whole
window on the index resident in HBM, result left in HBM: one launch of the sweep the library picks for the row format --
at the defaults sweep_conservation_halo3t_kernel<5, uint8, 256, false, true, true> (memo_sweep_cons3t.hip) reading the
k-class view of the dense rows.  Weak scaling: every rank owns its own 10^8-position window of an N x 10^8 pivot and
generates exactly the rows that window sees (index-addressable generator, memo_amd/synth.py); for N > 1 the result slices
are gathered to rank 0 over RCCL inside the timed region (memo_amd/bench_gather.py), and the line carries a short run of
BASELINE's multi-GPU configuration (config 5) next to its headline.

This file: argument parsing, the launcher, and the timed region, top to bottom.  The untimed legs -- the resident
indexes per row format with their one-time passes, the k-class view passes, the other row formats, the reference's seam
(memo_conservation), the CPU baseline -- are memo_amd/bench_legs.py; the N > 1 gather is memo_amd/bench_gather.py.

One JSON line on stdout (rank 0).  `roofline` prices the sweep kernel alone from HIP events recorded on the launch
stream; `cpu_baseline` is the oracle's literal port of the reference loop (memo_query.py:45-63,70) on one host core over
a bounded sample window.

Order of a run (N = 1): build the indexes (packing passes timed on the device by the library) -> view passes -> the same
query on the OTHER row formats (reported as other_row_formats; real work that also brings the clocks up) -> clock ramp:
headline launches until their time stops falling (untimed, bounded) -> W warm-up steps -> K timed steps -> K more steps
with an event pair each (per-step min / median, reported next to the mean) -> one_shot_seam -> CPU baseline + parity of
the timed result.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# numpy / torch / memo_amd are imported in main(), after launch_ranks() has had its say: the parent of a
# self-launched N > 1 run never loads the HIP runtime or the library.

WORKLOAD_NAMES = ("c2", "c3", "c4", "c5", "sparse", "c5h")     # memo_amd/bench_legs.py: WORKLOADS


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOAD_NAMES))
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


def cpu_baseline(R, out):
    """The oracle's literal transcription of memo_init + memo_query + argmax, one core, on the first `cpu_sample` positions
    of rank 0's window.  Also serves as the parity check of the benchmarked result (same positions).  The all-cores
    closed form runs on as many threads as the process's CPU budget pays for (allowed CPUs cut to the cgroup's quota:
    memo_host_threads) -- more would only be throttled."""
    import ctypes as C
    import numpy as np
    from memo_amd import synth
    from oracle import memo_oracle as oracle        # checker / baseline leg only
    args, num_docs, L, k, membership = R.args, R.num_docs, R.L, R.k, R.membership
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
    ok = bool(np.array_equal(want, R.result_as_host(out, S)))
    base = {"value": S / dt, "unit": "query-positions/s", "cores": 1, "kind": "port",
            "sample": f"first {S} positions of the window ({r1 - r0} rows), literal bool-matrix "
                      f"port of memo_query.py:45-63,70 (oracle_literal_*), {dt:.1f} s",
            "parity_with_gpu_on_sample": ok}
    # the strong CPU baseline of SURVEY.md 8(d)(ii): closed form (no L x N matrix), the window cut
    # into one sub-window per thread (the same sharding rule the GPUs use), all in C (GIL released)
    import concurrent.futures as cf
    allowed, quota = C.c_int32(), C.c_double()
    budget = R.lib.memo_host_threads(C.byref(allowed), C.byref(quota))
    threads = max(1, min(int(allowed.value) or (os.cpu_count() or 1), 64))
    if quota.value > 0:
        threads = max(1, min(threads, int(quota.value)))
    cuts = np.linspace(0, S, threads + 1).astype(np.int64)

    def part(i):
        a, b = int(cuts[i]), int(cuts[i + 1])
        i0 = int(np.searchsorted(s, a, side="right"))
        i1 = int(np.searchsorted(s, b + k, side="left"))
        return fn(s[i0:i1], e[i0:i1], o[i0:i1], a, b, k, num_docs, literal=False)
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(part, range(threads)))
    dt2 = time.perf_counter() - t0
    ok2 = bool(np.array_equal(np.concatenate(parts), want))
    base["all_cores"] = {"value": S / dt2, "unit": "query-positions/s", "cores": threads, "kind": "port",
                         "cpus_visible": os.cpu_count(), "cgroup_cpu_quota": float(quota.value) or None,
                         "sample": f"same {S} positions, closed-form restatement (oracle_closed_*) on {threads} "
                                   f"threads (the process's CPU budget: {allowed.value} CPUs allowed, cgroup quota "
                                   f"{quota.value:g}; the library's own pool would run {budget}), one sub-window each, {dt2:.2f} s",
                         "equal_to_literal_port": ok2}
    return base


def gathered_sample_vs_oracle(R, G, steps):
    """N > 1: the gathered slice of the LAST rank, checked against the oracle on a small sample; every gathered slice of the
    last step complete (bench_gather.Gather.last_rank_sample hands over the bytes; the oracle stays in this file)."""
    from memo_amd import synth
    import numpy as np
    from oracle import memo_oracle as oracle        # checker only
    g, a, got, whole = G.last_rank_sample(steps)
    S, k, num_docs = len(got), R.k, R.num_docs
    num, den = synth.rows_per_position(num_docs)
    sr0, sr1 = synth.shard_rows(a, a + S, k, num, den, R.pivot)
    s_, e_, o_ = oracle.synth_rows(sr0, sr1 - sr0, num, den, num_docs)
    want = (oracle.membership if R.membership else oracle.conservation)(s_, e_, o_, a, a + S, k, num_docs, literal=False)
    res = {"rank": g, "positions": S, "equal_to_oracle": bool(np.array_equal(got, want))}
    if whole is not None:
        res["every_slice_complete"] = whole
    return res


def main():
    args = parse()
    rc = launch_ranks(args, sys.argv[1:])
    if rc is not None:
        raise SystemExit(rc)
    import numpy as np
    import torch
    from memo_amd import bench_gather, bench_legs
    assert set(WORKLOAD_NAMES) == set(bench_legs.WORKLOADS)
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
    if os.environ.get("MEMO_BENCH_ONE_DEVICE") == "1":      # (the test transport: every rank on GPU 0, see bench_gather.Transport)
        local = 0
    torch.cuda.set_device(local)
    T = bench_gather.Transport(rank, world, torch.device("cuda", local), args.force_dist)
    multi = T.multi

    # ---- untimed: resident indexes, their one-time passes, the plan of the gather ---------------------------
    R = bench_legs.Run(args, rank, world, local)
    R.build_indexes(T.barrier if multi else None)
    R.result_layout()
    R.prepare_views()
    link_measured = ranks_seen = link_probe = None
    if multi:
        link_measured, ranks_seen, link_probe = bench_gather.probe_link(R, T)
    P = bench_gather.Plan(R, T, link_measured)              # coding of the slices, rank 0's share (N = 1: plain, whole)
    L, k = R.L, R.k
    share0 = max(8, int(L * P.root_weight) // 8 * 8)        # rank 0 sweeps the first root_weight of its window
    for o in R.outs:                                        # (every result buffer holds a whole-window result behind it)
        R.launch(o)
    if rank == 0:
        R.qe_mine = R.qs + (share0 if multi else L)
    positions_per_step = L * (world - 1) + (share0 if multi else L)
    G = bench_gather.Gather(R, T, P)
    if args.calibrate:
        R.indexes["wide"].debug_stream_rows(R.stream.cuda_stream)
    other = R.other_formats()                               # (1) the same query on the other row formats, for the record
    ramp = R.settle(lambda: R.launch(R.outs[0]))            # (2) settling on the headline kernel
    stream, ix = R.stream, R.ix

    for i in range(args.warmup):
        G.step(i)
    G.drain()
    ix.check(stream.cuda_stream)           # raises if the kernel flagged a bad row

    # ---- THE TIMED REGION: exactly K steps between barrier + synchronize on both sides ----------------------
    # HIP events on the launch stream: ONE pair around the K timed steps at N = 1 (the sweeps run back to back;
    # two event packets between every two launches cost the timed region 5-7 us per step) and none at all inside
    # the timed region at N > 1 (or --force-dist): there a step is sweep + coding + gather and the wall clock
    # between the barriers is what counts; the sweep alone is timed afterwards, K launches with a pair each.
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    if multi:
        T.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if not multi:
        ev[0].record(stream)
    for i in range(args.steps):
        G.step(i)
    if not multi:
        ev[1].record(stream)
    G.drain()                                # (torch.cuda.synchronize(): this rank's K steps are done, rank 0's incl. every slice received)
    dt = time.perf_counter() - t0            # N > 1: the MAX over ranks below is the barrier-to-barrier time of the job without
    if multi:                                # the closing barrier's own latency (~0.1 ms of RCCL against K = 20 steps of 0.2 ms)
        T.barrier()
    # ---- end of the timed region ------------------------------------------------------------------------------
    ix.check(stream.cuda_stream)
    out = R.outs[(args.steps - 1) % R.nbuf]
    # per-step spread (and, N > 1, the sweep's own time): K more sweeps with an event pair each
    per = R.per_step(lambda: R.launch(R.outs[args.steps % R.nbuf]), args.steps)
    kern_ms = float(np.mean(per)) if multi else ev[0].elapsed_time(ev[1]) / args.steps
    if multi:
        R.launch(R.outs[args.steps % R.nbuf])     # (leave every buffer holding a complete result)
        t = torch.tensor([dt, kern_ms], dtype=torch.float64, device=R.dev)
        T.all_reduce_max(t)
        dt, kern_ms = float(t[0]), float(t[1])

    res = None
    if rank == 0:
        # bytes of the row layout the timed kernel reads + the result it writes
        b_alg = R.row_bytes * R.rows_read[args.rows] + R.b_out * L
        res = {
            "metric": "query-positions/sec (chr window, k=%d)" % k,
            "value": positions_per_step * args.steps / dt,
            "unit": "query-positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64" if args.rows == "wide" else "u32",
            "data": "synthetic" if R.host_rows is None else "synthetic sequences (random pivot + mutated copies), real matching statistics",
            "config": bench_legs.config_entry(R, positions_per_step, P.root_weight, ramp),
            "roofline": bench_legs.roofline_entry(R, kern_ms, per, b_alg),
        }
        bench_legs.attach_traffic(R, res, b_alg)
        if multi:
            # what the same run delivers when the result slices stay on their GPUs (no root):
            # every rank's sweep time from its own HIP events, slowest rank counts
            res["without_gather"] = {"value": positions_per_step / (kern_ms * 1e-3), "unit": "query-positions/s",
                                     "note": "aggregate of the per-rank sweeps alone (max kernel_ms over ranks); "
                                             "`value` above includes delivering every slice to rank 0, which is "
                                             f"bound by {G.wires[0].numel() / 1e6:.0f} MB per peer link per step"}
            res["expected_from_model"] = P.expected(R, kern_ms)
        if other:
            res["other_row_formats"] = other
            wv = bench_legs.without_view_entry(R, other, kern_ms)
            if wv:
                res["resident_index_without_view"] = wv
        if world == 1 and args.cpu_sample > 0 and R.others and "wide" in R.indexes and R.host_rows is None and not R.membership:
            res["one_shot_seam"] = bench_legs.one_shot_seam(R, out)
        if world == 1 and args.cpu_sample > 0:
            res["cpu_baseline"] = cpu_baseline(R, out)
    if multi:
        if rank == 0:
            res["gather_parity_sample"] = gathered_sample_vs_oracle(R, G, args.steps)
            res["config"]["gather_payload"] = G.payload_note()
            if P.choice:
                P.choice["measured_step_ms"] = dt / args.steps * 1e3
                res["config"]["gather_coding_choice"] = P.choice
            from memo_amd import shard
            res["link_GBs_measured"] = link_measured / 1e9 if link_measured else None
            res["link_GBs_assumed_when_unmeasured"] = shard.XGMI_LINK_BYTES_PER_S / 1e9
            res["link_probe"] = link_probe
            res["ranks_seen"] = ranks_seen
            if T.staged:
                res["test_transport"] = ("gloo through host memory" + (", every rank on GPU 0" if os.environ.get("MEMO_BENCH_ONE_DEVICE") == "1" else "") +
                                         ": a run of the N > 1 control flow, NOT a bench line (value, link and step times mean nothing)")
    R.close()

    def write_line():
        if rank == 0:
            os.write(json_fd, (json.dumps(res) + "\n").encode())
    if world > 1 and args.workload != "c5":
        # BASELINE's multi-GPU configuration next to the headline: every rank takes part (bench_gather.config5_leg).  A leg
        # after the timed region must not cost the run its line: an exception is recorded, and a leg that does not come back
        # (a rank lost in a collective) is cut off by a timer on EVERY rank -- rank 0 writes the line it has, all exit 0.
        import threading
        limit = float(os.environ.get("MEMO_BENCH_CONFIG5_LIMIT_S", "240"))

        def cut_off():
            if rank == 0:
                res["config5"] = {"error": f"the config-5 leg did not finish in {limit:g} s"}
                write_line()
            os._exit(0)
        timer = threading.Timer(limit, cut_off)
        timer.daemon = True
        timer.start()
        try:
            c5 = bench_gather.config5_leg(args, rank, world, local, T, link_measured, gathered_sample_vs_oracle)
        except Exception as exc:
            c5 = {"error": repr(exc)}
        timer.cancel()
        if rank == 0:
            res["config5"] = c5
    write_line()
    T.close()


if __name__ == "__main__":
    main()
