"""Synthetic pangenome workloads (BASELINE.json configs 2-5; DESIGN.md "Synthetic workload").

Row i of the generator:  start = 1 + floor(i*den/num)   (num/den = rows per pivot position)
                         end   = start + mix(seed, 2i) % 60
                         annot = 1 + mix(seed, 2i+1) % (num_docs - 1)
The rows are produced on the device (memo_synth_fill); oracle_synth_rows is the CPU twin
used by the tests.  Index-addressable, so every shard generates exactly its own rows.
"""
from fractions import Fraction

from .index import DeviceIndex

SEED = 0x4D454D4F  # "MEMO"

# name -> (num_docs, pivot length, density = rows per genome-position)
CONFIGS = {
    "c2": dict(num_docs=10, pivot=10_000_000, density=Fraction(5, 100)),     # ~5 M rows
    "c3": dict(num_docs=100, pivot=100_000_000, density=Fraction(5, 100)),   # ~500 M rows
    "c4": dict(num_docs=100, pivot=100_000_000, density=Fraction(5, 100)),   # membership on c3
    "c5": dict(num_docs=500, pivot=3_000_000_000, density=Fraction(5, 100)),  # HPRC scale
}


def rows_per_position(num_docs, density=Fraction(5, 100)):
    f = Fraction(density) * num_docs
    return f.numerator, f.denominator


def first_row_at_or_after(x, num, den):
    """smallest i with start_i >= x, start_i = 1 + floor(i*den/num)."""
    if x <= 1:
        return 0
    return -((-(x - 1) * num) // den)          # ceil((x-1)*num/den)


def shard_rows(qs, qe, k, num, den, pivot):
    """global row range [r0, r1) of the rows a window [qs, qe) can see: qs < start < qe + k
    (memo_query.py:25-27 with :100), clipped to the pivot's total row count."""
    total = first_row_at_or_after(pivot, num, den)     # rows with start < pivot length
    r0 = min(first_row_at_or_after(qs + 1, num, den), total)
    r1 = min(first_row_at_or_after(qe + k, num, den), total)
    return r0, max(r1, r0)


def device_index(qs, qe, k, num_docs, pivot, density=Fraction(5, 100), device=0, seed=SEED, pack=None):
    """DeviceIndex holding exactly the rows window [qs, qe) needs (generated in HBM).
    pack: None = int64 columns only; "keep" = also the packed rows; "only" = packed rows only;
    "dense" = the dense rows only (memo_index_pack_dense; int64 columns and 4-byte rows dropped)."""
    num, den = rows_per_position(num_docs, density)
    r0, r1 = shard_rows(qs, qe, k, num, den, pivot)
    ix = DeviceIndex.synthetic(r1 - r0, r0, num, den, num_docs, seed=seed, device=device)
    if pack:
        ix.pack(keep_wide=(pack == "keep"))
    if pack == "dense":
        ix.pack_dense(keep_packed=False)
    return ix, (r0, r1)


def host_rows(row_begin, count, num, den, num_docs, seed=SEED):
    """the same rows as memo_synth_fill / oracle_synth_rows, generated with NumPy on the host
    (used to write synthetic Parquet indexes for end-to-end CLI runs)."""
    import numpy as np
    M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

    def mix(x):
        with np.errstate(over="ignore"):
            z = (np.uint64(seed) + (x + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) & M64
            z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
            z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
            return z ^ (z >> np.uint64(31))
    i = np.arange(row_begin, row_begin + count, dtype=np.uint64)
    start = (np.uint64(1) + (i * np.uint64(den)) // np.uint64(num)).astype(np.int64)
    end = start + (mix(np.uint64(2) * i) % np.uint64(60)).astype(np.int64)
    annot = (np.uint64(1) + mix(np.uint64(2) * i + np.uint64(1)) % np.uint64(num_docs - 1)).astype(np.int64)
    return start, end, annot


def write_parquet(path, num_docs, pivot, record="chr1", density=Fraction(5, 100), seed=SEED, chunk=4_000_000):
    """a synthetic index in the reference's on-disk format (f0 utf8, f1..f3 int64, ZSTD;
    parquet_compress_bed.py:19-38), written chunk by chunk."""
    import pyarrow as pa
    import pyarrow.parquet as pq
    num, den = rows_per_position(num_docs, density)
    total = first_row_at_or_after(pivot, num, den)
    schema = pa.schema([("f0", pa.utf8()), ("f1", pa.int64()), ("f2", pa.int64()), ("f3", pa.int64())])
    with pq.ParquetWriter(path, schema, compression="ZSTD") as w:
        for b in range(0, total, chunk):
            n = min(chunk, total - b)
            s, e, a = host_rows(b, n, num, den, num_docs, seed)
            w.write_table(pa.table({"f0": pa.array([record] * n, pa.utf8()), "f1": s, "f2": e, "f3": a}, schema=schema))
    return total
