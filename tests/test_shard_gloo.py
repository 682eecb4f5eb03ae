"""The N > 1 path on CPU: world_size-2 gloo processes shard a window, each runs its
sub-window (the oracle stands in for the sweep kernel here -- tests may use it as the
checker), and the gathered result equals the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from memo_amd import shard
from tests import golden_util as G


def test_split_window_covers_exactly():
    for qs, qe, world in ((0, 100, 2), (5, 6, 8), (17, 1000003, 8), (0, 0, 4), (3, 67, 3), (0, 16, 2)):
        wins, per = shard.split_window(qs, qe, world)
        assert len(wins) == world and wins[0][0] == qs and wins[-1][1] == max(qe, qs)
        assert all(a <= b for a, b in wins) and all(wins[i][1] == wins[i + 1][0] for i in range(world - 1))
        assert all(b - a <= per for a, b in wins) and per % shard.ALIGN == 0
        assert sum(b - a for a, b in wins) == max(qe - qs, 0)


def test_rows_for_window_is_the_reference_filter(oracle):
    s, e, o = G.index_columns("rnd_n40.parquet", "chr1")
    for a, b, k in ((0, 100, 31), (777, 1500, 3), (4000, 6000, 101), (4999, 5000, 31)):
        i0, i1 = shard.rows_for_window(s, a, b, k)
        fs, _, _ = oracle.filter_rows(s, e, o, a, b, k)
        assert np.array_equal(s[i0:i1], fs[fs > a])


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, membership, ret, root_weight=1.0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import memo_oracle as oracle
    s, e, o = G.index_columns("rnd_n40.parquet", "chr1")
    qs, qe, k, n = 123, 4321, 31, 40
    W = (n + 31) // 32

    def alloc(per):
        return torch.zeros((per, W), dtype=torch.int32) if membership else torch.zeros(per, dtype=torch.int16)

    def sweep(a, b, out):
        i0, i1 = shard.rows_for_window(s, a, b, k)
        rows = (s[i0:i1], e[i0:i1], o[i0:i1])
        if membership:
            r = oracle.membership(*rows, a, b, k, n, literal=False).view(np.int32)
        else:
            r = oracle.conservation(*rows, a, b, k, n, literal=False).view(np.int16)
        out[:b - a] = torch.from_numpy(r)

    res, _ = shard.sharded_query(sweep, qs, qe, k, rank, world, dist, alloc, root_weight=root_weight)
    if rank == 0:
        full = oracle.membership(s, e, o, qs, qe, k, n, literal=False).view(np.int32) if membership else \
            oracle.conservation(s, e, o, qs, qe, k, n, literal=False).view(np.int16)
        ret.put(bool(np.array_equal(res.numpy(), full)))
    dist.destroy_process_group()


@pytest.mark.parametrize("membership", [False, True])
@pytest.mark.parametrize("world,root_weight", [(2, 1.0), (3, 1.0), (3, 0.4), (2, 0.0)])
def test_gloo_sharded_equals_whole(world, root_weight, membership):
    """equal parts, a lighter part for rank 0 (it also gathers and decodes), and a root that only gathers;
    the partition is memo_split_window of the C ABI"""
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, membership, ret, root_weight)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get() is True
