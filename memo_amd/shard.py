"""Window sharding across ranks (one process per GPU) and the gather of result slices.

The reference is single-process; this is new.  It rests on one
property of the path: position p is influenced only by rows with
p < start <= p + k - 1, so a window [qs, qe) can be cut into contiguous sub-windows and each
sub-window queried on its own with the rows  a < start < b + k  -- the reference's own
filter (memo_query.py:25-27 with :100) applied to the sub-window.  No exchange during the
sweep; one gather of disjoint slices to the root at the end (RCCL over xGMI on GPUs:
every peer has a direct link to the root, so the G-1 sends run concurrently).
"""
import numpy as np

ALIGN = 8      # positions; keeps every uint16 slice 16-byte aligned in the gathered buffer


def split_window(qs, qe, world, align=ALIGN, root_weight=1.0):
    """[(a_g, b_g)] for g in range(world), and the longest part's length `per` (a multiple of `align`):
    contiguous parts, rank 0 gets `root_weight` shares of the window and every other rank one share
    (1.0 = equal parts; less when rank 0 also gathers and decodes the peers' slices), the tail takes
    what is left, ranks past the end get empty windows (a == b).  The rule itself is
    memo_split_window() of the C ABI (include/memo_amd_multi.h), which the single-process multi-GPU
    entry points use too; pure host arithmetic, no GPU needed."""
    import ctypes as C
    from ._lib import check, lib
    cuts = (C.c_int64 * (world + 1))()
    check(lib().memo_split_window(qs, qe, world, align, float(root_weight), cuts))
    wins = [(cuts[g], cuts[g + 1]) for g in range(world)]
    per = max(b - a for a, b in wins)
    per = -(-per // align) * align
    return wins, per


def rows_for_window(start_sorted, a, b, k):
    """index range [i0, i1) of the rows a sub-window sees: a < start < b + k."""
    i0 = int(np.searchsorted(start_sorted, a, side="right"))
    i1 = int(np.searchsorted(start_sorted, b + k, side="left"))
    return i0, max(i1, i0)


def gather_slices(local, per, lengths, rank, world, dist, dst=0):
    """local: this rank's slice, first dim padded to `per`.  Returns on dst the concatenation
    of the first lengths[g] entries of every slice (torch tensor), None elsewhere."""
    import torch
    if world == 1:
        return local[:lengths[0]]
    # slices travel as bytes: neither RCCL nor gloo carries 16-bit integers
    wire = local.contiguous().view(torch.uint8).reshape(-1)
    bufs = [torch.empty_like(wire) for _ in range(world)] if rank == dst else None
    dist.gather(wire, bufs, dst=dst)
    if rank != dst:
        return None
    parts = [bufs[g].view(local.dtype).reshape(local.shape)[:lengths[g]] for g in range(world)]
    return torch.cat(parts, dim=0)


def sharded_query(sweep, qs, qe, k, rank, world, dist, alloc, dst=0, root_weight=1.0):
    """Run `sweep(a, b, out)` on this rank's sub-window [a, b) of [qs, qe), writing into
    out[:b-a] (out = alloc(per)), then gather.  Returns (result on dst | None, (a, b))."""
    wins, per = split_window(qs, qe, world, root_weight=root_weight)
    a, b = wins[rank]
    out = alloc(per)
    if b > a:
        sweep(a, b, out)
    res = gather_slices(out, per, [w[1] - w[0] for w in wins], rank, world, dist, dst)
    return res, (a, b)


# ---- what a result slice should travel as (bench.py, N > 1) ---------------------------------------
XGMI_LINK_BYTES_PER_S = 75e9   # one direction of one xGMI link (7 links x ~153 GB/s per GPU, both directions summed)


def modelled_step(world, t_sweep, wire_bytes, t_decode, t_encode, t_decode_all=None, link=XGMI_LINK_BYTES_PER_S, root_weight=1.0):
    """seconds per step when every peer's slice goes to rank 0 coded: the slowest of a peer (sweep +
    encode), rank 0 (its own sweep of `root_weight` of a share + decoding the world - 1 slices it received;
    its own slice never travels) and a peer's link to rank 0 (each peer has its own; gather i overlaps
    sweep i + 1).  t_decode: one slice by a launch of its own; t_decode_all: the world - 1 slices of a step by ONE
    launch (memo_transport_runs_unpack_many_dev), where the coding has that form."""
    if world == 1:
        return t_sweep
    decode = t_decode_all if t_decode_all is not None else (world - 1) * t_decode
    return max(t_sweep + t_encode, root_weight * t_sweep + decode, wire_bytes / link)


def pick_coding(world, t_sweep, usable, link=XGMI_LINK_BYTES_PER_S):
    """usable: coding name -> (wire bytes, decode seconds per slice, encode seconds[, seconds to decode the world - 1 slices of a
    step in one launch]).  Returns the name
    with the shortest modelled step (ties: alphabetical, so that every rank would agree) and the model."""
    model = {c: modelled_step(world, t_sweep, *usable[c], link=link) for c in usable}
    return min(sorted(model), key=model.get), model


ROOT_WEIGHTS = tuple(round(1.0 - 0.05 * i, 2) for i in range(20))     # 1.0, 0.95, ..., 0.05


def pick_plan(world, t_sweep, usable, link=XGMI_LINK_BYTES_PER_S, weights=ROOT_WEIGHTS):
    """Coding AND root weight: rank 0 may sweep a smaller share of positions than its peers (it also decodes
    world - 1 slices per step).  Returns (coding, root_weight, modelled positions-per-second in units of one
    peer's window per second) for the plan with the highest modelled throughput (world - 1 + w) / step; ties go
    to the larger weight, then alphabetically, so that every rank would agree."""
    best = None
    for w in weights:
        for c in sorted(usable):
            step = modelled_step(world, t_sweep, *usable[c], link=link, root_weight=w)
            rate = (world - 1 + w) / step
            if best is None or rate > best[2] * (1 + 1e-9):
                best = (c, w, rate)
    return best
