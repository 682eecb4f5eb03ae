"""Sidecar cache of the packed rows of a chromosome, next to the Parquet index.

`memo query` spends its wall clock decoding ZSTD Parquet pages (0.59 s of 0.93 s for a 2 * 10^7-position
window of a 10^8-row index, profiles/r01_cli_timing_16decoders.txt); the sweep is 2 % of it.  The Parquet
file stays the source of truth.  Beside it, `<index>.parquet.memo/<record>.v3.pk` keeps, per record
(chromosome), exactly what the GPU wants: the packed rows (4 B per row; + 2 B when an annot exceeds 4095),
the dense rows (3.2 B per row, when every annot fits 9 bits: the benchmarked kernel's format),
the start-bucket table and the few rows with end < start -- as produced by the library itself
(memo_builder_* + memo_index_export_packed).  A repeat query maps the file, cuts the window's rows out with
two lookups in the bucket table and uploads them through the pinned ring (memo_index_import_packed): no
Parquet, no pyarrow import, no decode.

Validity: the header records the Parquet file's size and mtime_ns; any mismatch (or a different format
version, or a truncated file) makes the cache invisible and it is rebuilt.  Files are written under a
temporary name and renamed, so a reader never sees a partial file.  Set MEMO_CACHE=0 to neither read nor
write caches; MEMO_CACHE=read to read but never build.  A miss starts `python -m memo_amd.cache build`
detached in the background (the query itself is answered from the Parquet file as before); MEMO_CACHE=sync
builds it in-process after the answer is written instead (tests, batch jobs).  One builder per record at a
time (an O_EXCL lock file); a record whose rows cannot be packed leaves a marker (`.v3.nocache`, keyed like the
cache) so that later queries neither rebuild nor respawn.

v3 (round 4): beside the dense rows the file holds the k-class VIEW of them for the `-k` the cache was built for (the rows
whose overlap is below the class's cap: all a conservation query with k - 1 <= cap can be touched by -- what a resident index
builds once its class's queries have made it worth the pass, memo_index_info_t.last_rows_read) when that spares a fifth of the
dense rows: a hit with such a k uploads and sweeps fewer rows (a cached `memo query` spends half of its wall clock on slice +
upload).  Round 5: the view is the one the device builds (memo_index_export_view), not a host-side rebuild.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

from ._lib import MemoError, MemoUnpackable, check, lib
from .index import DeviceIndex, dense_rows_can_answer

VERSION = 3
HEADER_BYTES = 4096
MAGIC = b"MEMOPK03"
LOCK_STALE_SECONDS = 3600.0


def mode():
    v = os.environ.get("MEMO_CACHE", "1").lower()
    return {"0": "off", "off": "off", "no": "off", "read": "read", "sync": "sync"}.get(v, "on")


def _safe(record):
    """file-name form of a record name; injective ('_' is escaped too: 'chr 1' and 'chr_201' stay apart)"""
    return "".join(ch if (ch.isalnum() or ch in ".-") else "_%02x" % ord(ch) for ch in record) or "_"


def cache_path(in_file, record):
    return os.path.join(in_file + ".memo", _safe(record) + ".v%d.pk" % VERSION)


def _marker_path(in_file, record):      # "this record cannot be cached" (unpackable rows, or none)
    return os.path.join(in_file + ".memo", _safe(record) + ".v%d.nocache" % VERSION)


def _lock_path(in_file, record):
    return os.path.join(in_file + ".memo", _safe(record) + ".v%d.lock" % VERSION)


def _source_key(in_file):
    st = os.stat(in_file)
    return {"size": st.st_size, "mtime_ns": st.st_mtime_ns}


def _align(x, a=4096):
    return (x + a - 1) // a * a


def header_ok(head, file_bytes):
    """every offset / count of a header lies inside the file it describes (a corrupt or hand-edited header must make
    the cache invisible, not send raw pointers past the mapping)"""
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


def view_cap(k):
    """the cap of the k class of the dense rows (classes of two: overlaps below 2, 4 ... 32), or None when k has no class"""
    km1 = int(k) - 1
    return 2 * ((km1 + 1) // 2) if 1 <= km1 <= 32 else None


def write(in_file, record, ix, k=None):
    """ix: a packed DeviceIndex holding EVERY row of `record`.  Writes the cache file atomically: the 4- (6-) byte
    rows, the bucket table, the rows with end < start and -- when every annot fits 9 bits -- the DENSE rows too (3.2 B
    per row, memo_index_pack_dense: what the conservation sweep reads fastest, so that a cached `memo query -k 31`
    runs the benchmarked kernel)."""
    inf = ix.info()
    rows, nb, n_long, fmt = inf["rows"], inf["buckets"], inf["long_rows"], inf["packed_format"]
    if fmt not in (4, 6, 12) or inf["bucket_base"] != 0:
        raise ValueError("only a whole, packed chromosome can be cached")
    pk = np.empty(rows, np.uint32)
    pa = np.empty(rows if fmt == 6 else 0, np.uint16)
    boff = np.empty(nb, np.int64)
    longs = np.empty(3 * n_long, np.int64)
    check(lib().memo_index_export_packed(ix._h, pk.ctypes.data, pa.ctypes.data if fmt == 6 else None, boff.ctypes.data,
                                         longs.ctypes.data if n_long else None))
    p3, boff3, rows3 = np.empty(0, np.uint32), np.empty(0, np.int64), 0
    if (fmt == 4 or (fmt == 12 and inf["max_annot"] <= 511)) and rows:   # (annots of up to nine bits fit the dense rows)
        ix.pack_dense(keep_packed=True)
        rows3 = ix.info()["dense_row_count"]          # (fewer than rows when the rows that never write at k <= 64 were left out)
        p3 = np.empty(4 * ((rows3 + 4) // 5), np.uint32)
        boff3 = np.empty(nb, np.int64)                # the dense rows' own bucket table
        check(lib().memo_index_export_dense(ix._h, p3.ctypes.data, boff3.ctypes.data, longs.ctypes.data if n_long else None))
    view, pv, boffv = None, np.empty(0, np.uint32), np.empty(0, np.int64)
    cap = view_cap(k) if k is not None else None
    if cap and rows3:
        # the k class's view as the DEVICE builds it (memo_index_prepare + memo_index_export_view: a few ms, and the rows' places
        # inside their groups chosen against LDS bank conflicts) -- rounds 3-4 rebuilt it here with whole-chromosome NumPy
        # temporaries, 30-40 bytes of host memory per row.  Groups of five rows: what memo_index_import_dense takes.
        # The view is optional: whatever goes wrong building or fetching it (no room on the device, any HIP error) means "no
        # view in the file", never a failed cache write; and the caller's index gets its own MEMO_OPT_VIEW_ROWS back.
        made = None
        before = ix.set_option(4, 5)                  # MEMO_OPT_VIEW_ROWS
        try:
            ix.prepare(k, int(min(max(inf["max_annot"] + 1, 2), 511)))
            made = ix.export_view(k, 5)
        except MemoError:
            made = None
        finally:
            ix.set_option(4, before)
        if made is not None and made[3] == cap:
            pv, boffv, rows_v = made[0], made[1], made[2]
            view = {"cap": cap, "rows": rows_v}
    off_pk = HEADER_BYTES
    off_pa = _align(off_pk + pk.nbytes)
    off_p3 = _align(off_pa + pa.nbytes)
    off_boff3 = _align(off_p3 + p3.nbytes)
    off_pv = _align(off_boff3 + boff3.nbytes)
    off_boffv = _align(off_pv + pv.nbytes)
    off_boff = _align(off_boffv + boffv.nbytes)
    off_long = _align(off_boff + boff.nbytes)
    total = off_long + longs.nbytes
    if view:
        view.update(off_p3=off_pv, off_boff=off_boffv)
    head = {"version": VERSION, "record": record, "source": _source_key(in_file), "rows": rows, "format": fmt,
            "bucket_shift": inf["bucket_shift"], "buckets": nb, "min_start": inf["min_start"], "max_start": inf["max_start"],
            "max_annot": inf["max_annot"], "long_rows": n_long, "off_pk": off_pk, "off_pa": off_pa,
            "off_p3": off_p3 if p3.nbytes else None, "off_boff3": off_boff3, "rows3": rows3, "view": view, "off_boff": off_boff,
            "off_long": off_long, "bytes": total}
    blob = MAGIC + json.dumps(head).encode()
    if len(blob) > HEADER_BYTES:
        raise ValueError("cache header too large")
    path = cache_path(in_file, record)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "wb") as fh:
        fh.write(blob.ljust(HEADER_BYTES, b"\0"))
        for off, arr in ((off_pk, pk), (off_pa, pa), (off_p3, p3), (off_boff3, boff3), (off_pv, pv), (off_boffv, boffv), (off_boff, boff),
                         (off_long, longs)):
            fh.seek(off)
            fh.write(memoryview(arr).cast("B"))
        fh.truncate(total)
    os.replace(tmp, path)
    return path


def _open(in_file, record):
    """(header, memory map) of a valid cache file, else None"""
    path = cache_path(in_file, record)
    try:
        with open(path, "rb") as fh:
            raw = fh.read(HEADER_BYTES)
        if not raw.startswith(MAGIC):
            return None
        head = json.loads(raw[len(MAGIC):].rstrip(b"\0"))
        if head.get("version") != VERSION or head.get("record") != record or head.get("source") != _source_key(in_file):
            return None
        if not header_ok(head, os.path.getsize(path)):
            return None
        return head, np.memmap(path, dtype=np.uint8, mode="r")
    except (OSError, ValueError, KeyError):
        return None


def bucket_slice(head, entry, query_start, query_end, rows=None):
    """(b_lo, b_hi, r0, r1) of the rows with query_start < start < query_end (+ the rest of the two edge buckets);
    entry(b) reads table entry b (of the rows' own table: `rows` of them); None when the table is inconsistent"""
    nb, shift = head["buckets"], head["bucket_shift"]
    rows = head["rows"] if rows is None else rows
    b_lo = min(max(query_start, 0) >> shift, nb - 1)
    b_hi = min(max((max(query_end, 0) >> shift) + 1, b_lo), nb - 1)
    r0, r1 = int(entry(b_lo)), int(entry(b_hi))
    if not (0 <= r0 <= r1 <= rows):
        return None
    return b_lo, b_hi, r0, r1


def load_region(in_file, record, query_start, query_end, device=0, k=None, num_docs=None, membership=None):
    """DeviceIndex with the rows of `record` that have query_start < start < query_end (and a few more from
    the two buckets at the edges, which the sweep ignores as it ignores every row outside the window), from
    the cache; None when there is no valid cache.  With the query known (k, num_docs, membership) and dense rows in
    the file, the slice is imported as dense rows when they alone can answer it (memo_dense_rows_can_answer)."""
    got = _open(in_file, record)
    if got is None:
        return None
    head, mm = got
    nb, shift = head["buckets"], head["bucket_shift"]
    boff = mm[head["off_boff"]:head["off_boff"] + 8 * nb].view(np.int64)
    cut = bucket_slice(head, lambda b: boff[b], query_start, query_end)
    if cut is None:
        return None
    b_lo, b_hi, r0, r1 = cut
    table = boff[b_lo:b_hi + 1]                      # absolute entries: the library rebases them to r0
    n_long = head["long_rows"]
    longs = np.array(mm[head["off_long"]:head["off_long"] + 24 * n_long].view(np.int64)) if n_long else None
    h = C.c_void_p()
    n = r1 - r0
    lo, hi = slice_extent(head, b_lo, b_hi, n)
    if head.get("off_p3") is not None and k is not None and num_docs is not None:
        # the dense rows have their own numbering and table (rows that never write at k <= 64 may be left out of them)
        boff3 = mm[head["off_boff3"]:head["off_boff3"] + 8 * nb].view(np.int64)
        cut3 = bucket_slice(head, lambda b: boff3[b], query_start, query_end, rows=head["rows3"])
        if cut3 is not None and dense_rows_can_answer(cut3[3] - cut3[2], lo, hi, head["max_annot"], k, num_docs, membership):
            _, _, d0, d1 = cut3
            off_g, table_all = head["off_p3"], boff3
            view = head.get("view")
            if view is not None and k - 1 <= view["cap"]:          # this k's class of the dense rows: fewer rows to upload and sweep
                boffv = mm[view["off_boff"]:view["off_boff"] + 8 * nb].view(np.int64)
                cutv = bucket_slice(head, lambda b: boffv[b], query_start, query_end, rows=view["rows"])
                if cutv is not None and dense_rows_can_answer(cutv[3] - cutv[2], lo, hi, head["max_annot"], k, num_docs, membership):
                    _, _, d0, d1 = cutv
                    off_g, table_all = view["off_p3"], boffv
            base = d0 // 5 * 5                       # the slice starts with the group that holds row d0
            g = mm[off_g + 16 * (base // 5):off_g + 16 * ((d1 + 4) // 5)]
            table3 = table_all[b_lo:b_hi + 1]
            check(lib().memo_index_import_dense(d1 - base, device, shift, b_lo, g.ctypes.data, table3.ctypes.data, len(table3) + 1,
                                                base, lo, hi, head["max_annot"], longs.ctypes.data if n_long else None, n_long,
                                                C.byref(h)))
            return DeviceIndex(d1 - base, device, _handle=h)
    pk = mm[head["off_pk"] + 4 * r0:head["off_pk"] + 4 * r1]
    pa = mm[head["off_pa"] + 2 * r0:head["off_pa"] + 2 * r1] if head["format"] == 6 else None
    check(lib().memo_index_import_packed(n, device, shift, b_lo, pk.ctypes.data if n else None,
                                         pa.ctypes.data if (pa is not None and n) else (np.zeros(1, np.uint16).ctypes.data
                                                                                          if pa is not None else None),
                                         table.ctypes.data, len(table) + 1, r0, lo, hi, head["max_annot"],
                                         longs.ctypes.data if n_long else None, n_long, C.byref(h)))
    return DeviceIndex(n, device, _handle=h)


def slice_extent(head, b_lo, b_hi, n):
    """min / max start to give a slice of buckets b_lo .. b_hi (they are only used to tell dense from sparse
    indexes: the edges of the slice's buckets do)"""
    shift = head["bucket_shift"]
    if not n:
        return 0, -1
    return max(b_lo << shift, head["min_start"]), min(((b_hi + 1) << shift) - 1, head["max_start"])


# ---- building: at most one builder per record at a time, and a record that cannot be cached is remembered -------
def uncacheable(in_file, record):
    """a marker written by an earlier build says: this record (of this very Parquet file) cannot be cached"""
    try:
        with open(_marker_path(in_file, record)) as fh:
            return json.load(fh).get("source") == _source_key(in_file)
    except (OSError, ValueError):
        return False


def _mark_uncacheable(in_file, record, why):
    path = _marker_path(in_file, record)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "w") as fh:
        json.dump({"source": _source_key(in_file), "record": record, "why": why}, fh)
    os.replace(tmp, path)


def take_lock(in_file, record):
    """O_EXCL lock file: True when this process may build (or spawn the builder of) this record's cache.  A lock
    older than an hour is a builder that died; it is replaced."""
    path = _lock_path(in_file, record)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    for _ in range(2):
        try:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o644)
            os.write(fd, str(os.getpid()).encode())
            os.close(fd)
            return True
        except FileExistsError:
            try:
                if time.time() - os.stat(path).st_mtime < LOCK_STALE_SECONDS:
                    return False
                os.unlink(path)
            except OSError:
                return False
    return False


def release_lock(in_file, record):
    try:
        os.unlink(_lock_path(in_file, record))
    except OSError:
        pass


def build(in_file, record, device=0, locked=False, k=None):
    """decode every row of `record` from the Parquet file, pack it with the library (the host packer only: rows it
    refuses are NOT uploaded as int64 columns -- 24 B per row of a whole chromosome for a cache that cannot be
    written), write the cache; a record that cannot be cached gets a marker so that no later query tries again.
    locked: the caller (build_in_background's parent) already holds the record's lock."""
    from .memo_query import region_index
    if not locked and not take_lock(in_file, record):
        return None                         # somebody else is building it
    try:
        if _open(in_file, record) is not None:
            return cache_path(in_file, record)
        hi = (1 << 61) - 1
        try:
            ix = region_index(in_file, record, -1, hi, device=device, k=2, use_cache=False, packed_only=True)
        except MemoUnpackable as exc:
            _mark_uncacheable(in_file, record, str(exc))
            return None
        with ix:
            inf = ix.info()
            if inf["packed_format"] not in (4, 6, 12) or inf["rows"] == 0:
                _mark_uncacheable(in_file, record, "no rows" if inf["rows"] == 0 else "rows are not packed")
                return None
            return write(in_file, record, ix, k=k)
    finally:
        release_lock(in_file, record)


def build_in_background(in_file, record, k=None):
    """a detached `python -m memo_amd.cache build` -- unless the record is known to be uncacheable or somebody
    (this user's other queries) is already building it"""
    import subprocess
    if uncacheable(in_file, record) or not take_lock(in_file, record):
        return False
    env = dict(os.environ, MEMO_CACHE="read", MEMO_CACHE_LOCKED="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    try:
        if k is not None:
            env["MEMO_CACHE_K"] = str(int(k))     # (the query's k: its class's view of the dense rows goes into the file too)
        subprocess.Popen([sys.executable, "-m", "memo_amd.cache", "build", in_file, record], env=env,
                         stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                         start_new_session=True)
        return True
    except OSError:
        release_lock(in_file, record)
        return False


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "build":
        dev = int(os.environ.get("MEMO_DEVICE", "0"))
        held = bool(os.environ.get("MEMO_CACHE_LOCKED")) and len(sys.argv) == 4
        kk = int(os.environ["MEMO_CACHE_K"]) if os.environ.get("MEMO_CACHE_K", "").isdigit() else None
        for rec in sys.argv[3:]:
            print(build(sys.argv[2], rec, dev, locked=held, k=kk))
    else:
        sys.exit("usage: python -m memo_amd.cache build INDEX.parquet RECORD [RECORD ...]")
