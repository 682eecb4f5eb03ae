"""Python face of the CPU oracle (TEST INFRASTRUCTURE -- see memo_oracle.c header).

Two independent restatements of /root/reference/src/memo_query.py:42-71 live here:

* ``C``      ctypes binding of oracle/libmemo_oracle.so (literal bool-matrix
             transcription + closed forms + emitters + synthetic generator);
* ``np_*``   a NumPy restatement of the closed forms, written differently from
             both the reference and the C file (expand every surviving row into
             its (position, order) pairs, then ``np.minimum.at`` /
             ``np.bitwise_and.at``), so that a slip in one restatement is caught
             by the other.  Both are pinned against tests/golden/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmemo_oracle.so")

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "memo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.environ.get("MEMO_ORACLE_LIB") or build())   # override: the ASan/UBSan build (make asan)
        q = [_i64p, _i64p, _i64p, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_int64]
        for name, outp in (("oracle_literal_conservation", _u16p), ("oracle_closed_conservation", _u16p),
                           ("oracle_literal_membership", _u32p), ("oracle_closed_membership", _u32p)):
            f = getattr(L, name)
            f.argtypes = q + [outp]
            f.restype = C.c_int
        L.oracle_filter.argtypes = [_i64p, _i64p, _i64p, C.c_uint64, C.c_int64, C.c_int64, C.c_int64,
                                    _i64p, _i64p, _i64p]
        L.oracle_filter.restype = C.c_uint64
        L.oracle_emit_conservation.argtypes = [_u16p, C.c_int64, C.c_char_p]
        L.oracle_emit_conservation.restype = C.c_size_t
        L.oracle_emit_membership.argtypes = [_u32p, C.c_int64, C.c_int64, C.c_char_p]
        L.oracle_emit_membership.restype = C.c_size_t
        L.oracle_synth_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64,
                                        C.c_uint64, _i64p, _i64p, _i64p]
        L.oracle_synth_rows.restype = None
        L.oracle_fnv1a.argtypes = [_u8p, C.c_uint64]
        L.oracle_fnv1a.restype = C.c_uint64
        _lib = L
    return _lib


class OracleIndexError(IndexError):
    """The reference raises IndexError (NumPy) / is undefined (Numba) here."""


def _cols(s, e, o):
    return (np.ascontiguousarray(s, np.int64), np.ascontiguousarray(e, np.int64),
            np.ascontiguousarray(o, np.int64))


def filter_rows(s, e, o, qs, qe, k):
    """filter_pq (memo_query.py:19-36 with the +k of :100) on one chromosome's columns."""
    s, e, o = _cols(s, e, o)
    fs, fe, fo = np.empty_like(s), np.empty_like(s), np.empty_like(s)
    n = lib().oracle_filter(s, e, o, len(s), qs, qe, k, fs, fe, fo)
    return fs[:n].copy(), fe[:n].copy(), fo[:n].copy()


def _run(fn, s, e, o, qs, qe, k, n_docs, out):
    s, e, o = _cols(s, e, o)
    rc = fn(s, e, o, len(s), qs, qe, k, n_docs, out)
    if rc == -1:
        raise OracleIndexError("order column outside the result matrix")
    if rc == -3:
        raise ValueError("negative dimensions are not allowed")     # np.zeros([qe - qs, ...])
    if rc != 0:
        raise MemoryError("oracle: cannot allocate the L x N matrix")
    return out


def conservation(s, e, o, qs, qe, k, n_docs, literal=True):
    out = np.empty(max(qe - qs, 0), np.uint16)
    fn = lib().oracle_literal_conservation if literal else lib().oracle_closed_conservation
    return _run(fn, s, e, o, qs, qe, k, n_docs, out)


def membership(s, e, o, qs, qe, k, n_docs, literal=True):
    out = np.empty(max(qe - qs, 0) * ((n_docs + 31) // 32), np.uint32)
    fn = lib().oracle_literal_membership if literal else lib().oracle_closed_membership
    return _run(fn, s, e, o, qs, qe, k, n_docs, out).reshape(max(qe - qs, 0), (n_docs + 31) // 32)


def emit_conservation(vec):
    vec = np.ascontiguousarray(vec, np.uint16)
    buf = C.create_string_buffer(6 * len(vec) + 2)
    n = lib().oracle_emit_conservation(vec, len(vec), buf)
    return buf.raw[:n]


def emit_membership(bits, n_docs):
    bits = np.ascontiguousarray(bits, np.uint32)
    L = bits.shape[0] if bits.ndim == 2 else (len(bits) // max((n_docs + 31) // 32, 1))
    buf = C.create_string_buffer(2 * n_docs * L + 2)
    n = lib().oracle_emit_membership(bits.reshape(-1), L, n_docs, buf)
    return buf.raw[:n]


def synth_rows(row_begin, count, num, den, n_docs, seed=0x4D454D4F):
    s, e, o = (np.empty(count, np.int64) for _ in range(3))
    lib().oracle_synth_rows(row_begin, count, num, den, n_docs, seed, s, e, o)
    return s, e, o


def synth_window_compare(got, qs, qe, k, n_docs, pivot, membership=False, threads=None, chunk=2_000_000,
                         density=(5, 100)):
    """Whole-window check of a result on the synthetic index (memo_amd/synth.py): the window is cut into
    chunks, every chunk regenerates the rows it sees (qs' < start < qe' + k) and runs the closed-form
    restatement, one chunk per thread (the C calls release the GIL).  `got` is the device result for
    [qs, qe) (uint8 / uint16 vector, or uint32 [L, W] bit rows).  Returns (number of chunks that
    differ, FNV-1a of the concatenated per-chunk FNV-1a values of the ORACLE's result) -- the second
    is a checksum of checksums that two runs of any implementation can be compared by."""
    import concurrent.futures as cf
    from fractions import Fraction
    f = Fraction(*density) * n_docs
    num, den = f.numerator, f.denominator

    def first_row_at_or_after(x):
        return 0 if x <= 1 else -((-(x - 1) * num) // den)
    total = first_row_at_or_after(pivot)
    threads = threads or max(1, min(os.cpu_count() or 1, 64))
    cuts = list(range(qs, qe, chunk)) + [qe]
    is_memb = bool(membership)

    def part(i):
        a, b = cuts[i], cuts[i + 1]
        r0 = min(first_row_at_or_after(a + 1), total)
        r1 = max(min(first_row_at_or_after(b + k), total), r0)
        s, e, o = synth_rows(r0, r1 - r0, num, den, n_docs)
        want = (globals()["membership"] if is_memb else conservation)(s, e, o, a, b, k, n_docs, literal=False)
        mine = got[a - qs:b - qs]
        same = np.array_equal(mine.astype(want.dtype) if not is_memb else mine, want)
        return (0 if same else 1), fnv1a(want)
    with cf.ThreadPoolExecutor(threads) as pool:
        res = list(pool.map(part, range(len(cuts) - 1)))
    bad = sum(r[0] for r in res)
    return bad, fnv1a(np.array([r[1] for r in res], np.uint64))


def fnv1a(arr):
    b = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    return int(lib().oracle_fnv1a(b, b.size))


# ---------------------------------------------------------------------------
# NumPy restatement of the closed forms (independent of the C file)
# ---------------------------------------------------------------------------
def _covered_pairs(s, e, o, qs, qe, k, ncols):
    """(position, column) pairs written by memo_query.py:61-62 after :45-49."""
    L = qe - qs
    if L < 0:
        raise ValueError("negative dimensions are not allowed")
    s = np.asarray(s, np.int64)
    e = np.asarray(e, np.int64)
    o = np.asarray(o, np.int64)
    hi = np.minimum(np.maximum(s - qs, 0), L)
    lo = np.minimum(np.maximum(e - qs - (k - 1), 0), L)
    live = lo < hi
    lo, hi, col = lo[live], hi[live], o[live]
    col = np.where(col < 0, col + ncols, col)
    if col.size and (col.min() < 0 or col.max() >= ncols):
        raise OracleIndexError("order column outside the result matrix")
    span = hi - lo
    owner = np.repeat(np.arange(span.size), span)
    pos = np.arange(span.sum()) - np.repeat(np.cumsum(span) - span, span) + lo[owner]
    return pos, col[owner]


def np_conservation(s, e, o, qs, qe, k, n_docs):
    out = np.full(max(qe - qs, 0), n_docs, np.int64)
    pos, col = _covered_pairs(s, e, o, qs, qe, k, n_docs + 1)
    np.minimum.at(out, pos, col)
    return out.astype(np.uint16)


def np_membership(s, e, o, qs, qe, k, n_docs):
    L, W = max(qe - qs, 0), (n_docs + 31) // 32
    full = np.zeros(W, np.uint32)
    for g in range(n_docs):
        full[g >> 5] |= np.uint32(1) << np.uint32(g & 31)
    out = np.tile(full, (L, 1))
    pos, col = _covered_pairs(s, e, o, qs, qe, k, n_docs)
    np.bitwise_and.at(out, (pos, col >> 5), ~(np.uint32(1) << (col & 31).astype(np.uint32)))
    return out


def bits_to_matrix(bits, n_docs):
    """uint32 [L, W] bit rows -> uint8 [L, N] like the reference's rec.astype('byte')."""
    bits = np.asarray(bits, np.uint32)
    g = np.arange(n_docs)
    return ((bits[:, g >> 5] >> (g & 31).astype(np.uint32)) & 1).astype(np.uint8)


# ---------------------------------------------------------------------------
# `memo view` preprocessing (plot_conservation.py:46-65), restated with NumPy
# ---------------------------------------------------------------------------
def view_table(vec, n_docs, n_bins):
    """(bin, No. Genomes, value) rows as the reference's melted DataFrame holds them."""
    vec = np.asarray(vec, np.int64)
    edges = [int(x) for x in np.linspace(0, len(vec), n_bins + 1)]          # :52
    value = np.zeros((n_bins, n_docs + 1))
    for b, (lo, hi) in enumerate(zip(edges[:-1], edges[1:])):
        if hi == lo:
            raise ZeroDivisionError("division by zero")                      # :56, empty Counter
        seg = vec[lo:hi]
        for order in range(n_docs + 1):
            value[b, order] = int((seg == order).sum()) / (hi - lo)
    keep = np.arange(n_docs)                                                 # :65 drops order n_docs
    return {"bin": np.tile(np.arange(n_bins, dtype=np.int64), n_docs),
            "No. Genomes": np.repeat(keep.astype(np.float64), n_bins),
            "value": value[:, :n_docs].T.reshape(-1)}
