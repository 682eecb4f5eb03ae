"""DeviceIndex: one chromosome of a MEMO index resident in HBM, and the sweep launches.

Counterpart of the arrays /root/reference/src/memo_query.py passes between filter_pq,
memo_init and memo_query (lines 28-36, 45-55): three int64 columns (start, end, annot).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def _col(a):
    a = np.ascontiguousarray(a)
    if a.dtype != np.int64:
        # filter_pq builds np.uint (memo_query.py:28-35) and memo_init re-types to int64 (:45)
        a = a.astype(np.int64) if a.dtype != np.uint64 else a.view(np.int64)
    return a


def _ptr(x):
    """device pointer of a torch tensor / raw int address."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


def words(num_docs):
    return (num_docs + 31) // 32


def dense_rows_can_answer(rows, min_start, max_start, max_annot, k, num_docs, membership):
    """memo_dense_rows_can_answer: would an index holding only the dense rows (3.2 B per row) answer this query?"""
    try:
        return bool(lib().memo_dense_rows_can_answer(int(rows), int(min_start), int(max_start), int(max_annot), int(k),
                                                     int(num_docs), 1 if membership else 0))
    except (C.ArgumentError, OverflowError, ValueError, TypeError):
        return False


class DeviceIndex:
    def __init__(self, rows, device=0, _handle=None):
        self._h = C.c_void_p()
        self.rows = int(rows)
        self.device = int(device)
        if _handle is not None:
            self._h = _handle
        else:
            check(lib().memo_index_create(self.rows, self.device, C.byref(self._h)))

    @classmethod
    def from_host_packed(cls, start, end, annot, device=0, bucket_shift=0, dense=False):
        """the packed, pinned way in (memo_builder_*): rows narrowed on the host, 4 B per row over PCIe -- or, with
        dense=True, 3.2 B per row as the dense rows the benchmarked conservation kernel reads (conservation, k <= 64,
        <= 255 genomes: dense_rows_can_answer) -- an index that is finalized and packed.  Raises MemoUnpackable
        for rows that need the next way in (dense -> 4-byte words -> from_host)."""
        s, e, o = _col(start), _col(end), _col(annot)
        if not (len(s) == len(e) == len(o)):
            raise ValueError("columns differ in length")
        with IndexBuilder(len(s), device, bucket_shift, dense=dense) as b:
            b.push(s, e, o)
            return b.finish()

    @classmethod
    def from_host(cls, start, end, annot, device=0, bucket_shift=0, allow_sort=True):
        s, e, o = _col(start), _col(end), _col(annot)
        if not (len(s) == len(e) == len(o)):
            raise ValueError("columns differ in length")
        ix = cls(len(s), device)
        check(lib().memo_index_upload(ix._h, s.ctypes.data, e.ctypes.data, o.ctypes.data, len(s)))
        ix.finalize(bucket_shift, allow_sort)
        return ix

    def upload_rows(self, offset, start, end, annot):
        s, e, o = _col(start), _col(end), _col(annot)
        check(lib().memo_index_upload_rows(self._h, offset, s.ctypes.data, e.ctypes.data, o.ctypes.data, len(s)))

    def truncate(self, rows):
        check(lib().memo_index_truncate(self._h, rows))
        self.rows = int(rows)

    @classmethod
    def synthetic(cls, rows, row_begin, num, den, num_docs, seed=0x4D454D4F, device=0, bucket_shift=0):
        ix = cls(rows, device)
        check(lib().memo_synth_fill(ix._h, row_begin, num, den, num_docs, seed))
        ix.finalize(bucket_shift, False)
        return ix

    def columns(self):
        s, e, o = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().memo_index_columns(self._h, C.byref(s), C.byref(e), C.byref(o)))
        return s.value, e.value, o.value

    def finalize(self, bucket_shift=0, allow_sort=True):
        check(lib().memo_index_finalize(self._h, bucket_shift, 1 if allow_sort else 0))
        return self

    def pack(self, keep_wide=True):
        """build the 4/6-byte-per-row query format (memo_index_pack)"""
        check(lib().memo_index_pack(self._h, 1 if keep_wide else 0))
        return self

    def pack_dense(self, keep_packed=True):
        """build the dense rows (memo_index_pack_dense: five rows per 16 bytes): k <= 64, annot <= 511 (above 255: conservation only, uint16)"""
        check(lib().memo_index_pack_dense(self._h, 1 if keep_packed else 0))
        return self

    def info(self):
        inf = _lib.IndexInfo()
        inf.struct_bytes = C.sizeof(inf)           # the versioned struct: the library writes no more than this
        check(lib().memo_index_get_info_v5(self._h, C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in inf._fields_}

    def prepare(self, k, num_docs, membership=False, window_hint=0, stream=None):
        """memo_index_prepare: build now what queries of this kind would build on the way (the k-class view of the rows,
        the tile table, the query order of rows that came in through the builder); returns the device bytes it took"""
        taken = C.c_uint64(0)
        check(lib().memo_index_prepare(self._h, int(k), int(num_docs), 1 if membership else 0, int(window_hint), _ptr(stream),
                                       C.byref(taken)))
        return taken.value

    def export_view(self, k, rows_per_group=5):
        """memo_index_export_view: (groups uint32[4 g], table int64[buckets], rows, cap) of the resident k-class view of the dense
        rows, or None when there is none (memo_index_prepare builds it)"""
        rows, ng, cap = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        check(lib().memo_index_export_view(self._h, int(k), int(rows_per_group), None, None, C.byref(rows), C.byref(ng), C.byref(cap)))
        if not rows.value:
            return None
        groups = np.empty(4 * ng.value, np.uint32)
        table = np.empty(self.info()["buckets"], np.int64)
        check(lib().memo_index_export_view(self._h, int(k), int(rows_per_group), groups.ctypes.data, table.ctypes.data, C.byref(rows),
                                           C.byref(ng), C.byref(cap)))
        return groups, table, rows.value, cap.value

    def set_option(self, option, value):
        """memo_index_set_option: 1 = MEMO_OPT_VIEWS (0 / 1), 2 = MEMO_OPT_VIEW_BUDGET_PCT, 3 = MEMO_OPT_BUILD_COST_PCT (100: ski rental; 0: the
        first query of a class builds), 4 = MEMO_OPT_VIEW_ROWS (0 / 5 / 6), 5 = MEMO_OPT_VIEW_PLACES (0 / 1).  Returns the option's previous value."""
        return check(lib().memo_index_set_option(self._h, int(option), int(value)))     # the option's previous value

    # ---- asynchronous launches on device buffers (torch tensors or raw addresses) ----
    def conservation_dev(self, qs, qe, k, num_docs, out, stream=None):
        check(lib().memo_query_conservation_dev(self._h, qs, qe, k, num_docs, _ptr(out), _ptr(stream)))

    def conservation_u8_dev(self, qs, qe, k, num_docs, out, stream=None):
        check(lib().memo_query_conservation_u8_dev(self._h, qs, qe, k, num_docs, _ptr(out), _ptr(stream)))

    def membership_dev(self, qs, qe, k, num_docs, out, stream=None):
        check(lib().memo_query_membership_dev(self._h, qs, qe, k, num_docs, _ptr(out), _ptr(stream)))

    # ---- include/memo_amd_debug.h: only with _lib.use_ab() (libmemo_amd_ab.so) ----
    def debug_stream_rows(self, stream=None):
        check(lib().memo_debug_stream_rows(self._h, _ptr(stream)))

    def debug_no_views(self, on=True):
        check(lib().memo_debug_no_views(self._h, 1 if on else 0))
        return self

    def debug_row_order(self, order):
        check(lib().memo_debug_row_order(self._h, int(order)))

    def debug_set_tuning(self, tile_w=0, waves=0, membership_algo=0, row_source=0, scatter=0):
        check(lib().memo_debug_set_tuning(self._h, tile_w, waves, membership_algo, row_source, scatter))
        return self

    def check(self, stream=None):
        check(lib().memo_query_check(self._h, _ptr(stream)))

    # ---- blocking host-result forms ----
    def _run(self, fn, qs, qe, k, num_docs, out):
        nbytes = out.nbytes
        d = C.c_void_p()
        check(lib().memo_dev_malloc(self.device, nbytes, C.byref(d)))
        try:
            check(fn(self._h, qs, qe, k, num_docs, d, None))
            self.check()
            check(lib().memo_dev_download(self.device, out.ctypes.data, d, nbytes, None))
        finally:
            lib().memo_dev_free(self.device, d)
        return out

    def conservation(self, qs, qe, k, num_docs, dtype=np.uint16):
        out = np.empty(max(qe - qs, 0), dtype)
        fn = lib().memo_query_conservation_dev if out.itemsize == 2 else lib().memo_query_conservation_u8_dev
        return self._run(fn, qs, qe, k, num_docs, out)

    def membership(self, qs, qe, k, num_docs):
        out = np.empty((max(qe - qs, 0), words(num_docs)), np.uint32)
        return self._run(lib().memo_query_membership_dev, qs, qe, k, num_docs, out)

    def close(self):
        if self._h:
            lib().memo_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class IndexBuilder:
    """memo_builder_*: start-sorted host rows -> a packed, finalized DeviceIndex, piece by piece."""

    def __init__(self, max_rows, device=0, bucket_shift=0, dense=False):
        self._b = C.c_void_p()
        self.device = int(device)
        self.rows = 0
        self.dense = bool(dense)
        check(lib().memo_builder_create_rows(int(max_rows), self.device, bucket_shift, 1 if dense else 0, C.byref(self._b)))

    def push(self, start, end, annot):
        s, e, o = _col(start), _col(end), _col(annot)
        check(lib().memo_builder_push(self._b, s.ctypes.data, e.ctypes.data, o.ctypes.data, len(s)))
        self.rows += len(s)

    def push_rows(self, rows):
        """memo_builder_push_rows: an [M, 3] row-major array (start, end, annot side by side) as it lies"""
        r = _rows3(rows)
        check(lib().memo_builder_push_rows(self._b, r.ctypes.data, len(r)))
        self.rows += len(r)

    def finish(self):
        h = C.c_void_p()
        check(lib().memo_builder_finish(self._b, C.byref(h)))
        return DeviceIndex(self.rows, self.device, _handle=h)

    def close(self):
        if self._b:
            lib().memo_builder_destroy(self._b)
            self._b = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- one-shot host API: memo_init + memo_query + reduction on host arrays ----
def conservation(start, end, annot, qs, qe, k, num_docs, device=0):
    s, e, o = _col(start), _col(end), _col(annot)
    out = np.empty(max(qe - qs, 0), np.uint16)
    check(lib().memo_conservation(s.ctypes.data, e.ctypes.data, o.ctypes.data, len(s), qs, qe, k,
                                  num_docs, out.ctypes.data, device))
    return out


def membership(start, end, annot, qs, qe, k, num_docs, device=0):
    s, e, o = _col(start), _col(end), _col(annot)
    out = np.empty((max(qe - qs, 0), words(num_docs)), np.uint32)
    check(lib().memo_membership(s.ctypes.data, e.ctypes.data, o.ctypes.data, len(s), qs, qe, k,
                                num_docs, out.ctypes.data, device))
    return out


def _rows3(rows):
    """filter_pq's array ([M, 3] uint64 / int64, row-major) as the ABI's rows3: no copy when it already is that"""
    a = np.asarray(rows)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("rows must be an [M, 3] array (start, end, annot)")
    if a.dtype != np.int64:
        a = a.view(np.int64) if a.dtype == np.uint64 else a.astype(np.int64)
    return np.ascontiguousarray(a)


def conservation_rows(rows, qs, qe, k, num_docs, device=0):
    """memo_conservation_rows: memo_init's first argument as it is (memo_query.py:103) in, the conservation vector out"""
    r = _rows3(rows)
    out = np.empty(max(qe - qs, 0), np.uint16)
    check(lib().memo_conservation_rows(r.ctypes.data, len(r), qs, qe, k, num_docs, out.ctypes.data, device))
    return out


def membership_rows(rows, qs, qe, k, num_docs, device=0):
    r = _rows3(rows)
    out = np.empty((max(qe - qs, 0), words(num_docs)), np.uint32)
    check(lib().memo_membership_rows(r.ctypes.data, len(r), qs, qe, k, num_docs, out.ctypes.data, device))
    return out


# ---- several GPUs from one process (include/memo_amd_multi.h) ----
def conservation_multi(start, end, annot, qs, qe, k, num_docs, devices):
    s, e, o = _col(start), _col(end), _col(annot)
    out = np.empty(max(qe - qs, 0), np.uint16)
    dev = (C.c_int32 * len(devices))(*devices)
    check(lib().memo_conservation_multi(s.ctypes.data, e.ctypes.data, o.ctypes.data, len(s), qs, qe, k, num_docs,
                                        out.ctypes.data, dev, len(devices)))
    return out


def membership_multi(start, end, annot, qs, qe, k, num_docs, devices):
    s, e, o = _col(start), _col(end), _col(annot)
    out = np.empty((max(qe - qs, 0), words(num_docs)), np.uint32)
    dev = (C.c_int32 * len(devices))(*devices)
    check(lib().memo_membership_multi(s.ctypes.data, e.ctypes.data, o.ctypes.data, len(s), qs, qe, k, num_docs,
                                      out.ctypes.data, dev, len(devices)))
    return out


def query_multi_dev(shards, qs, qe, k, num_docs, out, root_device=0, stream=None, root_weight=1.0, membership=False):
    """shards: DeviceIndex per device (replicas, or position shards covering their sub-windows); out: device
    buffer on root_device.  Asynchronous on `stream`; check() every shard afterwards."""
    hs = (C.c_void_p * len(shards))(*[ix._h for ix in shards])
    fn = lib().memo_query_membership_multi_dev if membership else lib().memo_query_conservation_multi_dev
    check(fn(hs, len(shards), qs, qe, k, num_docs, _ptr(out), root_device, _ptr(stream), float(root_weight)))


# ---- print_res text (memo_query.py:65-71) ----
def emit_conservation_buffer(vec):
    """the text as a uint8 array (no extra copies; write it with fh.write(memoryview(buf)))"""
    vec = np.ascontiguousarray(vec, np.uint16)
    need = lib().memo_emit_conservation(vec.ctypes.data, len(vec), None, 0)
    buf = np.empty(need, np.uint8)
    lib().memo_emit_conservation(vec.ctypes.data, len(vec), buf.ctypes.data, need)
    return buf


def emit_membership_buffer(bits, num_docs):
    bits = np.ascontiguousarray(bits, np.uint32)
    L = bits.shape[0] if bits.ndim == 2 else len(bits) // max(words(num_docs), 1)
    need = lib().memo_emit_membership(bits.ctypes.data, L, num_docs, None, 0)
    buf = np.empty(need, np.uint8)
    lib().memo_emit_membership(bits.ctypes.data, L, num_docs, buf.ctypes.data, need)
    return buf


def emit_conservation(vec):
    return emit_conservation_buffer(vec).tobytes()


def emit_membership(bits, num_docs):
    return emit_membership_buffer(bits, num_docs).tobytes()


def bits_to_matrix(bits, num_docs):
    """uint32 [L, W] -> uint8 [L, N], the reference's rec.astype('byte') (memo_query.py:68)."""
    bits = np.asarray(bits, np.uint32).reshape(-1, max(words(num_docs), 1))
    g = np.arange(num_docs)
    return ((bits[:, g >> 5] >> (g & 31).astype(np.uint32)) & 1).astype(np.uint8)
