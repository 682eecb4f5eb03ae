"""`memo query` answered from the sidecar cache with nothing but ctypes.

A repeat query of a cached record needs no Parquet, no Arrow and no NumPy: the packed rows are a memory-mapped
file (memo_amd/cache.py), the window's slice is two lookups in its bucket table, and everything else happens
behind the C ABI (memo_index_import_packed, memo_query_*_dev, memo_emit_*).  Importing NumPy alone costs as much
as all of that together, so this module imports only the standard library and memo_amd._lib; bin/memo tries it
first and falls back to memo_amd.memo_query.main (same results, same errors) whenever it does not apply: no valid
cache, k > 256, MEMO_CACHE=0, a sharded run, unparsable arguments.

Same output bytes as the reference's print_res (/root/reference/src/memo_query.py:65-71): the text comes from the
same C emitters the regular path uses.
"""
import ctypes as C
import json
import mmap
import os
import struct
import sys
import time

from . import _lib

VERSION = 3
HEADER_BYTES = 4096
MAGIC = b"MEMOPK03"


def _cache_path(in_file, record):       # (= memo_amd.cache.cache_path; restated here to keep NumPy out)
    safe = "".join(ch if (ch.isalnum() or ch in ".-") else "_%02x" % ord(ch) for ch in record) or "_"
    return os.path.join(in_file + ".memo", safe + ".v%d.pk" % VERSION)


def _header_ok(head, file_bytes):       # (= memo_amd.cache.header_ok)
    try:
        rows, nb, n_long, fmt = int(head["rows"]), int(head["buckets"]), int(head["long_rows"]), int(head["format"])
        total = int(head["bytes"])
        if total != file_bytes or rows < 0 or nb < 2 or n_long < 0 or fmt not in (4, 6, 12):
            return False
        if not (1 <= int(head["bucket_shift"]) <= 8) or int(head["max_annot"]) < 0:
            return False
        need = [(int(head["off_pk"]), 4 * rows), (int(head["off_boff"]), 8 * nb), (int(head["off_long"]), 24 * n_long)]
        if fmt == 6:
            need.append((int(head["off_pa"]), 2 * rows))
        if head.get("off_p3") is not None:
            rows3 = int(head["rows3"])
            if not (0 <= rows3 <= rows):
                return False
            need += [(int(head["off_p3"]), 16 * ((rows3 + 4) // 5)), (int(head["off_boff3"]), 8 * nb)]
        view = head.get("view")
        if view is not None:
            rows_v, cap = int(view["rows"]), int(view["cap"])
            if head.get("off_p3") is None or not (0 <= rows_v <= int(head["rows3"])) or not (2 <= cap <= 32):
                return False
            need += [(int(view["off_p3"]), 16 * ((rows_v + 4) // 5)), (int(view["off_boff"]), 8 * nb)]
        return all(off >= HEADER_BYTES and size >= 0 and off + size <= total for off, size in need)
    except (KeyError, TypeError, ValueError):
        return False


def _lazy_buffer(nbytes):
    """(mmap, address) of an anonymous mapping: pages appear when they are first written -- by the library's threads, not by a
    zero-fill on this one, which is what a ctypes array costs (203 MB of text + 200 MB of result at BASELINE config 3).  No
    MADV_HUGEPAGE: with it the faults of the emitter's threads took longer, not shorter (profiles/r06_cli_timing.txt)."""
    mm = mmap.mmap(-1, max(nbytes, 1))
    return mm, C.addressof(C.c_char.from_buffer(mm))


def _open(in_file, record):
    """(header, writable-copy memory map, file size) of a valid cache file, else None"""
    path = _cache_path(in_file, record)
    try:
        st = os.stat(in_file)
        with open(path, "rb") as fh:
            raw = fh.read(HEADER_BYTES)
            if not raw.startswith(MAGIC):
                return None
            head = json.loads(raw[len(MAGIC):].rstrip(b"\0"))
            if head.get("version") != VERSION or head.get("record") != record:
                return None
            if head.get("source") != {"size": st.st_size, "mtime_ns": st.st_mtime_ns}:
                return None
            if not _header_ok(head, os.fstat(fh.fileno()).st_size):
                return None
            # ACCESS_COPY: private, copy-on-write -- never written to, but ctypes wants a writable buffer
            return head, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_COPY)
    except (OSError, ValueError, KeyError):
        return None


def try_query(in_file, region, k, num_docs, out_file, membership):
    """True when the query was answered (out_file written) from the cache; False when this path does not
    apply and the caller should run the regular one.  Errors of the query itself (the reference's IndexError /
    ValueError cases) are raised exactly as the regular path raises them."""
    mode = os.environ.get("MEMO_CACHE", "1").lower()
    if mode in ("0", "off", "no") or os.environ.get("MEMO_QUERY_WIDE") or os.environ.get("MEMO_FORCE_SHARDED"):
        return False
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return False
    try:
        num_docs, k = int(num_docs), int(k)
        record, start_end = region.split(":")
        qs, qe = map(int, start_end.split("-"))
    except (ValueError, TypeError, AttributeError):
        return False                              # the regular path raises what the reference raises
    if not (1 < k <= 256) or not os.path.isfile(in_file):
        return False
    t0 = time.perf_counter()
    got = _open(in_file, record)
    if got is None:
        return False
    head, mm = got
    lib = _lib.lib()
    device = int(os.environ.get("MEMO_DEVICE", "0"))
    nb, shift = head["buckets"], head["bucket_shift"]

    def entry(b):
        return struct.unpack_from("<q", mm, head["off_boff"] + 8 * b)[0]
    q_end = qe + k                                # filter_pq's upper bound (memo_query.py:100)
    b_lo = min(max(qs, 0) >> shift, nb - 1)
    b_hi = min(max((max(q_end, 0) >> shift) + 1, b_lo), nb - 1)
    r0, r1 = entry(b_lo), entry(b_hi)
    if not (0 <= r0 <= r1 <= head["rows"]):
        return False                              # a table that does not describe these rows: the regular path answers
    n = r1 - r0
    base = C.addressof(C.c_char.from_buffer(mm))
    n_long = head["long_rows"]
    lo = max(b_lo << shift, head["min_start"]) if n else 0
    hi = min(((b_hi + 1) << shift) - 1, head["max_start"]) if n else -1
    ix = C.c_void_p()
    d_out = C.c_void_p()
    dense = False
    try:
        # the dense rows (3.2 B per row, the benchmarked conservation kernel) when the file has them and they alone can
        # answer this query; else the 4- / 6-byte rows
        d0 = d1 = 0
        if head.get("off_p3") is not None:        # (the dense rows' own numbering and bucket table)
            d0 = struct.unpack_from("<q", mm, head["off_boff3"] + 8 * b_lo)[0]
            d1 = struct.unpack_from("<q", mm, head["off_boff3"] + 8 * b_hi)[0]
            dense = 0 <= d0 <= d1 <= head["rows3"] and bool(lib.memo_dense_rows_can_answer(
                d1 - d0, lo, hi, head["max_annot"], k, num_docs, 1 if membership else 0))
        off_g, off_t, viewed = head.get("off_p3"), head.get("off_boff3"), False
        view = head.get("view")
        if dense and view is not None and k - 1 <= view["cap"]:
            # this k's class of the dense rows (the rows whose overlap is below the cap: all that can write at this k), when the
            # cache was built for such a k: fewer rows to upload and sweep
            v0 = struct.unpack_from("<q", mm, view["off_boff"] + 8 * b_lo)[0]
            v1 = struct.unpack_from("<q", mm, view["off_boff"] + 8 * b_hi)[0]
            if 0 <= v0 <= v1 <= view["rows"] and lib.memo_dense_rows_can_answer(v1 - v0, lo, hi, head["max_annot"], k, num_docs, 0):
                d0, d1, off_g, off_t, viewed = v0, v1, view["off_p3"], view["off_boff"], True
        if dense:
            row_base = d0 // 5 * 5                # the slice starts with the group that holds row d0
            n = d1 - d0
            _lib.check(lib.memo_index_import_dense(
                d1 - row_base, device, shift, b_lo, base + off_g + 16 * (row_base // 5),
                base + off_t + 8 * b_lo, b_hi - b_lo + 2, row_base, lo, hi, head["max_annot"],
                (base + head["off_long"]) if n_long else None, n_long, C.byref(ix)))
        else:
            _lib.check(lib.memo_index_import_packed(
                n, device, shift, b_lo, base + head["off_pk"] + 4 * r0,
                (base + head["off_pa"] + 2 * r0) if head["format"] == 6 else None,
                base + head["off_boff"] + 8 * b_lo, b_hi - b_lo + 2, r0, lo, hi, head["max_annot"],
                (base + head["off_long"]) if n_long else None, n_long, C.byref(ix)))
        t1 = time.perf_counter()
        L = max(qe - qs, 0)
        words = (num_docs + 31) // 32
        nbytes = 4 * L * words if membership else 2 * L
        host_mm, host = _lazy_buffer(nbytes)        # (a ctypes array would be zero-filled here, by this one thread)
        _lib.check(lib.memo_dev_malloc(device, max(nbytes, 16), C.byref(d_out)))
        fn = lib.memo_query_membership_dev if membership else lib.memo_query_conservation_dev
        _lib.check(fn(ix, qs, qe, k, num_docs, d_out, None))        # (qe < qs: the reference's ValueError)
        _lib.check(lib.memo_query_check(ix, None))                  # (annot outside the matrix: its IndexError)
        _lib.check(lib.memo_dev_download(device, host, d_out, nbytes, None))
        t2 = time.perf_counter()
        if membership:
            need = lib.memo_emit_membership(host, L, num_docs, None, 0)
        else:
            need = lib.memo_emit_conservation(host, L, None, 0)
        text_mm, text = _lazy_buffer(need)          # (its pages are first touched by the emitter's threads)
        if membership:
            lib.memo_emit_membership(host, L, num_docs, text, need)
        else:
            lib.memo_emit_conservation(host, L, text, need)
        with open(out_file, "wb") as fh:
            fh.write(memoryview(text_mm)[:need])
        t3 = time.perf_counter()
    finally:
        if d_out:
            lib.memo_dev_free(device, d_out)
        if ix:
            lib.memo_index_destroy(ix)
    if os.environ.get("MEMO_TIMING"):
        sys.stderr.write("memo_query timing: region slice+upload %.3f s (from the sidecar cache, ctypes-only path), "
                         "sweep+download %.3f s, text+write %.3f s (%d rows as %s, %d positions)\n"
                         % (t1 - t0, t2 - t1, t3 - t2, n, ("dense rows (3.2 B)" + (", the k-class view" if viewed else "")) if dense
                            else "4-byte rows", L))
    return True
