#!/usr/bin/env python3
"""Host side of `memo query` -- same command line, same function names and argument
meaning as the reference's src/memo_query.py, with the NumPy/Numba work replaced by
the HIP sweep behind include/memo_amd.h.

  reference (src/memo_query.py)            here
  ---------------------------------------  ------------------------------------------------
  filter_pq            :19-36              filter_pq      Arrow region slice -> SoA int64
  memo_init            :42-55              memo_init      upload + finalize a DeviceIndex,
                                                          describe the result buffer
  memo_query (@jit)    :57-63              memo_query     launch the sweep kernel, download
  print_res            :65-71              print_res      byte-identical text
  parse_arguments/main :76-105             parse_arguments / main

Run:  memo_query.py -b bed.parquet -r chr:start-end -k K -n N -o out.txt [-m]
The GPU is chosen with the MEMO_DEVICE environment variable (default 0), so the flag set
stays the reference's.
"""
import argparse
import os

import numpy as np

from . import _lib
from ._lib import MemoUnpackable
from .index import (DeviceIndex, IndexBuilder, dense_rows_can_answer, emit_conservation_buffer, emit_membership_buffer,
                    words)


class RegionRows:
    """Rows of one region as three int64 columns (what the Parquet file stores,
    parquet_compress_bed.py:21-26).  ``as_array()`` gives the reference's uint64 [M, 3]."""

    def __init__(self, start, end, annot):
        self.start, self.end, self.annot = start, end, annot

    def __len__(self):
        return len(self.start)

    def as_array(self):
        return np.stack([self.start, self.end, self.annot], axis=1).astype(np.uint64)


def _columns(table):
    return [np.ascontiguousarray(table.column(c).to_numpy(), dtype=np.int64) for c in ("f1", "f2", "f3")]


def filter_pq(in_file, query_record, query_start, query_end, spanning_rows=False):
    """Rows of `query_record` with query_start < f1 < query_end  (memo_query.py:25-27; main
    passes query_end + k, :100).  Row groups are pruned by the f0/f1 statistics the index
    writer leaves in the file; columns come back as Arrow buffers, no pandas, no AoS copy.

    The reference also fetches rows with f1 <= query_start < f2 (:22-24).  memo_init drops
    every one of them (their start clips to 0, so casted_end < start cannot hold, :46-49);
    they are read only when spanning_rows=True, which reproduces the reference's return
    value row for row (spanning rows first)."""
    import pyarrow.dataset as ds          # pulls in pandas: kept off the CLI's path (region_index)
    pq_ds = ds.dataset(in_file, format="parquet")
    rec = ds.field("f0") == query_record
    inside = pq_ds.to_table(filter=rec & (ds.field("f1") > query_start) & (ds.field("f1") < query_end),
                            columns=["f1", "f2", "f3"])
    cols = _columns(inside)
    if spanning_rows:
        span = pq_ds.to_table(filter=rec & (ds.field("f1") <= query_start) & (ds.field("f2") > query_start),
                              columns=["f1", "f2", "f3"])
        cols = [np.concatenate([a, b]) for a, b in zip(_columns(span), cols)]
    return RegionRows(*cols)


def region_chunks(in_file, query_record, query_start, query_end):
    """(upper bound on rows, iterator of (start, end, annot) chunks) for the rows of
    `query_record` with query_start < f1 < query_end in ONE Parquet file.  Row groups are selected
    by their f0 / f1 min-max statistics and decoded by a small thread pool, a bounded window ahead
    of the consumer, in file order."""
    import concurrent.futures as cf
    import pyarrow.compute as pc
    import pyarrow.parquet as pq
    pf = pq.ParquetFile(in_file)
    md = pf.metadata
    names = [md.schema.column(i).name for i in range(md.num_columns)]
    i0, i1 = names.index("f0"), names.index("f1")
    groups, bound = [], 0
    one_record, inside = {}, {}
    for g in range(md.num_row_groups):
        rg = md.row_group(g)
        s0, s1 = rg.column(i0).statistics, rg.column(i1).statistics
        if s0 is not None and s0.has_min_max and not (s0.min <= query_record <= s0.max):
            continue
        if s1 is not None and s1.has_min_max and not (s1.max > query_start and s1.min < query_end):
            continue
        groups.append(g)
        bound += rg.num_rows
        # what the statistics already prove about every row of the group: f0 is the record (then the name column --
        # two thirds of the decode -- is not read), f1 lies inside the window (then nothing is filtered)
        one_record[g] = bool(s0 is not None and s0.has_min_max and s0.min == query_record == s0.max and
                             s0.has_null_count and s0.null_count == 0)
        inside[g] = bool(s1 is not None and s1.has_min_max and s1.min > query_start and s1.max < query_end and
                         s1.has_null_count and s1.null_count == 0)

    import threading
    tls = threading.local()

    def load(g):
        if not hasattr(tls, "pf"):              # one reader per decode thread: a ParquetFile's
            tls.pf = pq.ParquetFile(in_file)    # read cache is not safe to share
        t = tls.pf.read_row_group(g, columns=["f1", "f2", "f3"] if one_record[g] else ["f0", "f1", "f2", "f3"])
        if one_record[g] and inside[g]:
            return _columns(t)
        f1 = t.column("f1")
        keep = pc.and_(pc.greater(f1, query_start), pc.less(f1, query_end))
        if not one_record[g]:
            keep = pc.and_(pc.equal(t.column("f0"), query_record), keep)
        return _columns(t.filter(keep))

    def chunks():
        # row groups decode in parallel (Arrow releases the GIL), a bounded window ahead of the
        # consumer, and are handed over in file order so that the rows stay start-sorted
        # (as many decoders as the library would run host threads: the CPUs allowed, cut to the cgroup's CFS quota -- memo_host_threads)
        depth = max(1, int(os.environ.get("MEMO_DECODE_THREADS") or _lib.lib().memo_host_threads(None, None)))
        with cf.ThreadPoolExecutor(max_workers=depth) as pool:
            window = [pool.submit(load, g) for g in groups[:depth]]
            for n in range(len(groups)):
                cols = window.pop(0).result()
                if n + depth < len(groups):
                    window.append(pool.submit(load, groups[n + depth]))
                if len(cols[0]):
                    yield cols
    return bound, chunks()


def region_index(in_file, query_record, query_start, query_end, device=None, k=None, use_cache=True,
                 num_docs=None, membership=None, packed_only=False):
    """filter_pq + the upload half of memo_init in one streaming pass: the rows go from the
    Parquet file straight into a finalized DeviceIndex, row group by row group (the next one is
    decoded by Arrow while the current one is packed and copied to the GPU); the host never holds more
    than a few row groups.  For k <= 256 (k=None: unknown, as large as it likes) the rows take the packed
    way in (memo_builder_*: narrowed on the host into pinned memory, PackedRows kernels) -- as DENSE rows
    (3.2 B per row, the benchmarked conservation kernel) when the query is known (num_docs, membership) and
    the dense rows alone can answer it (memo_dense_rows_can_answer), else as 4-byte words; rows that cannot be
    packed, and larger k, go up as int64 columns (packed_only: raise MemoUnpackable instead).  Anything that
    is not a single Parquet file goes through filter_pq."""
    device = _device() if device is None else device
    packed = k is not None and 1 < k <= 256 and not os.environ.get("MEMO_QUERY_WIDE")
    hint = (k, num_docs, bool(membership)) if (packed and num_docs is not None and not os.environ.get("MEMO_QUERY_PACKED")) else None
    miss = None
    if packed and use_cache and os.path.isfile(in_file):
        from . import cache                       # sidecar cache of the packed rows (memo_amd/cache.py)
        if cache.mode() != "off":
            index = cache.load_region(in_file, query_record, query_start, query_end, device, *(hint or (None, None, None)))
            if index is not None:
                index.cache = "hit"
                return index
            miss = (in_file, query_record)
    index = _region_index_from_parquet(in_file, query_record, query_start, query_end, device, packed, hint, packed_only)
    index.cache = miss                            # main() builds the cache after the answer is written
    return index


def _dense_first(rows, lo, hi, max_annot, hint):
    """[True, False] = try the dense rows, then the 4-byte words; [False] = words only"""
    if hint is None:
        return [False]
    k, num_docs, membership = hint
    return [True, False] if dense_rows_can_answer(rows, lo, hi, max_annot, k, num_docs, membership) else [False]


def _region_index_from_parquet(in_file, query_record, query_start, query_end, device, packed, hint=None, packed_only=False):
    if not os.path.isfile(in_file):
        rows = filter_pq(in_file, query_record, query_start, query_end)
        if packed:
            index = _from_host_rows(rows.start, rows.end, rows.annot, device, hint)
            if index is not None:
                return index
            if packed_only:
                raise MemoUnpackable(-6, "rows cannot be packed")
        return DeviceIndex.from_host(rows.start, rows.end, rows.annot, device=device)
    if packed:
        # density judged from the row groups' statistics before anything is decoded: `bound` rows (an upper bound)
        # between query_start and query_end; the finished index is asked again (it knows its rows and largest annot)
        bound, chunks = region_chunks(in_file, query_record, query_start, query_end)
        for dense in _dense_first(bound, max(query_start, 0), max(query_end, 0), 0, hint):
            if chunks is None:
                bound, chunks = region_chunks(in_file, query_record, query_start, query_end)
            try:
                with IndexBuilder(bound, device, dense=dense) as builder:
                    for cols in chunks:
                        builder.push(*cols)
                    index = builder.finish()
                if dense:
                    inf = index.info()
                    if not dense_rows_can_answer(inf["rows"], inf["min_start"], inf["max_start"], inf["max_annot"], *hint):
                        index.close()             # (sparser than the statistics promised, or an annot outside the matrix)
                        chunks = None
                        continue
                return index
            except MemoUnpackable:
                chunks = None                     # read the slice again: 4-byte words, then int64 columns
        if packed_only:
            raise MemoUnpackable(-6, "rows of %s cannot take the packed way in" % query_record)
    bound, chunks = region_chunks(in_file, query_record, query_start, query_end)
    index = DeviceIndex(bound, device)
    written = 0
    for cols in chunks:
        index.upload_rows(written, *cols)
        written += len(cols[0])
    index.truncate(written)
    return index.finalize()


def _from_host_rows(s, e, o, device, hint):
    """host columns -> packed index: dense rows when they can answer the query, else 4-byte words; None = unpackable"""
    s = np.asarray(s)
    n = len(s)
    tries = [False]
    if hint is not None and n:
        o_ = np.asarray(o)
        tries = _dense_first(n, int(s[0]), int(s[-1]), int(o_.max()) if int(o_.min()) >= 0 else 1 << 20, hint)
    for dense in tries:
        try:
            return DeviceIndex.from_host_packed(s, e, o, device=device, dense=dense)
        except MemoUnpackable:
            continue
    return None


class QueryResult:
    """Counterpart of the reference's `rec` matrix, kept in reduced form:
    conservation -> uint16 [L] (= argmax over columns, :70); membership -> uint32 [L, W] bits."""

    def __init__(self, true_start, true_end, k, num_docs, membership_query):
        self.true_start, self.true_end, self.k = true_start, true_end, k
        self.num_docs, self.membership_query = num_docs, membership_query
        self.values = None

    @property
    def true_len(self):
        return max(self.true_end - self.true_start, 0)


def _device():
    return int(os.environ.get("MEMO_DEVICE", "0"))


def memo_init(mem_arr, k, true_start, true_end, num_docs, membership_query):
    """Put the rows in HBM (validated, start-bucketed) and describe the result.
    Recentring, the k-1 shadow cast, clipping and the casted_end < start filter of
    memo_query.py:45-49 happen inside the sweep kernel, per tile."""
    if isinstance(mem_arr, DeviceIndex):      # region_index() already put the rows in HBM
        return mem_arr, QueryResult(true_start, true_end, k, num_docs, membership_query)
    if isinstance(mem_arr, RegionRows):
        s, e, o = mem_arr.start, mem_arr.end, mem_arr.annot
    else:                                     # the reference's [M, 3] array
        arr = np.asarray(mem_arr)
        s, e, o = (np.ascontiguousarray(arr[:, i]) for i in range(3))
    index = None
    if 1 < k <= 256:
        hint = None if os.environ.get("MEMO_QUERY_PACKED") else (k, num_docs, bool(membership_query))
        index = _from_host_rows(s, e, o, _device(), hint)
    if index is None:
        index = DeviceIndex.from_host(s, e, o, device=_device())
    return index, QueryResult(true_start, true_end, k, num_docs, membership_query)


def memo_query(mem_arr, rec, membership_query):
    """Run the conservation (min order) or membership (per-genome bit) sweep."""
    if membership_query:
        rec.values = mem_arr.membership(rec.true_start, rec.true_end, rec.k, rec.num_docs)
    else:
        rec.values = mem_arr.conservation(rec.true_start, rec.true_end, rec.k, rec.num_docs)
    return rec


def print_res(rec, out_file, membership_query):
    """Same bytes as memo_query.py:65-71."""
    text = emit_membership_buffer(rec.values, rec.num_docs) if membership_query else \
        emit_conservation_buffer(rec.values)
    with open(out_file, "wb") as fh:
        fh.write(memoryview(text))


################################################################################

def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description="Extract and query overlap MEMs for k-mer presence/absence.")
    parser.add_argument('-b', '--pq_bed_file', dest='in_file', help='parquet bed file', required=True)
    parser.add_argument('-o', '--out_file', dest='out_file', help='output file', required=True)
    parser.add_argument('-n', '--ndocs', dest='num_docs', help='total number of genomes in the pangenome', required=True)
    parser.add_argument('-k', '--kmer_size', dest='k', help='k-mer size', required=True)
    parser.add_argument('-r', '--genome_region', dest='genome_region', help='genome region, formatted as chr:start-end', required=True)
    parser.add_argument('-m', '--membership_query', dest='membership_query', action='store_true', default=False,
                        help='Perform membership query instead of conservation query')
    return parser.parse_args(argv)


def _main_sharded(args):
    """`memo query` under torch.distributed.run: one process per GPU, the window cut into
    contiguous sub-windows (memo_amd/shard.py), every rank slices + sweeps its own, result slices
    gathered to rank 0 over RCCL, rank 0 writes the file.  Same bytes as the single-GPU run."""
    import torch
    import torch.distributed as dist
    from . import shard
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    membership_query = args.membership_query
    num_docs, k = int(args.num_docs), int(args.k)
    query_record, start_end = args.genome_region.split(':')
    query_start, query_end = map(int, start_end.split('-'))
    if query_end < query_start:
        raise ValueError("negative dimensions are not allowed")          # np.zeros of memo_init
    stream = torch.cuda.current_stream()
    held = []

    def alloc(per):
        return torch.empty((per, words(num_docs)) if membership_query else (per,),
                           dtype=torch.int32 if membership_query else torch.int16, device=dev)

    def sweep(a, b, out):
        ix = region_index(args.in_file, query_record, a, b + k, device=local, k=k, num_docs=num_docs,
                          membership=membership_query)
        held.append(ix)
        if membership_query:
            ix.membership_dev(a, b, k, num_docs, out, stream.cuda_stream)
        else:
            ix.conservation_dev(a, b, k, num_docs, out, stream.cuda_stream)
        ix.check(stream.cuda_stream)

    try:
        res, _ = shard.sharded_query(sweep, query_start, query_end, k, rank, world, dist, alloc)
    finally:
        for ix in held:
            ix.close()
    if rank == 0:
        rec = QueryResult(query_start, query_end, k, num_docs, membership_query)
        host = res.cpu().numpy()
        rec.values = host.view(np.uint32) if membership_query else host.view(np.uint16)
        print_res(rec, args.out_file, membership_query)
    dist.barrier()
    dist.destroy_process_group()


def main(args):
    import sys
    import time
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("MEMO_FORCE_SHARDED"):
        return _main_sharded(args)
    membership_query = args.membership_query
    num_docs, k = int(args.num_docs), int(args.k)
    query_record, start_end = args.genome_region.split(':')      # exactly one ':' and one '-'
    query_start, query_end = map(int, start_end.split('-'))
    t = [time.perf_counter()]
    rows = region_index(args.in_file, query_record, query_start, query_end + k, k=k,      # filter_pq, :100
                        num_docs=num_docs, membership=membership_query)
    t.append(time.perf_counter())
    mem_arr, rec = memo_init(rows, k, query_start, query_end, num_docs, membership_query)
    try:
        rec = memo_query(mem_arr, rec, membership_query)
        inf = mem_arr.info() if os.environ.get("MEMO_TIMING") else None
    finally:
        mem_arr.close()
    t.append(time.perf_counter())
    print_res(rec, args.out_file, membership_query)
    t.append(time.perf_counter())
    miss = getattr(rows, "cache", None)
    if isinstance(miss, tuple) and mem_arr.rows:          # answered from the Parquet file: leave a cache for next time
        from . import cache
        try:                                              # the answer is written: a cache that cannot be left
            if cache.uncacheable(*miss):                  # (read-only directory, full disk) is not an error
                pass                                      # an earlier build found this record's rows unpackable
            elif cache.mode() == "sync":
                cache.build(*miss, device=_device(), k=None if membership_query else k)
            elif cache.mode() == "on":
                cache.build_in_background(*miss, k=None if membership_query else k)
        except (OSError, RuntimeError) as exc:
            if os.environ.get("MEMO_TIMING"):
                sys.stderr.write("memo_query: no sidecar cache written (%s)\n" % exc)
    if os.environ.get("MEMO_TIMING"):          # stderr only: stdout stays the reference's
        fmt = ("int64 columns" if inf["has_wide"] else "dense rows (3.2 B)" if inf["dense_rows"] else
               "%d-byte rows" % (6 if inf["packed_format"] == 6 else 4))
        sys.stderr.write("memo_query timing: region slice+upload %.3f s%s, sweep+download %.3f s, text+write %.3f s "
                         "(%d rows as %s, kernel family %d, %d positions)\n"
                         % (t[1] - t[0], " (from the sidecar cache)" if miss == "hit" else "", t[2] - t[1], t[3] - t[2],
                            mem_arr.rows, fmt, inf["last_sweep"], max(query_end - query_start, 0)))


if __name__ == "__main__":
    main(parse_arguments())
