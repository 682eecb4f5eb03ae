"""Sidecar cache of the packed rows of a chromosome, next to the Parquet index.

`memo query` spends its wall clock decoding ZSTD Parquet pages (0.59 s of 0.93 s for a 2 * 10^7-position
window of a 10^8-row index, profiles/r01_cli_timing_16decoders.txt); the sweep is 2 % of it.  The Parquet
file stays the source of truth.  Beside it, `<index>.parquet.memo/<record>.v1.pk` keeps, per record
(chromosome), exactly what the GPU wants: the packed rows (4 B per row; + 2 B when an annot exceeds 4095),
the start-bucket table and the few rows with end < start -- as produced by the library itself
(memo_builder_* + memo_index_export_packed).  A repeat query maps the file, cuts the window's rows out with
two lookups in the bucket table and uploads them through the pinned ring (memo_index_import_packed): no
Parquet, no pyarrow import, no decode.

Validity: the header records the Parquet file's size and mtime_ns; any mismatch (or a different format
version, or a truncated file) makes the cache invisible and it is rebuilt.  Files are written under a
temporary name and renamed, so a reader never sees a partial file.  Set MEMO_CACHE=0 to neither read nor
write caches; MEMO_CACHE=read to read but never build.  A miss starts `python -m memo_amd.cache build`
detached in the background (the query itself is answered from the Parquet file as before); MEMO_CACHE=sync
builds it in-process after the answer is written instead (tests, batch jobs).
"""
import ctypes as C
import json
import os
import sys

import numpy as np

from ._lib import check, lib
from .index import DeviceIndex

VERSION = 1
HEADER_BYTES = 4096
MAGIC = b"MEMOPK01"


def mode():
    v = os.environ.get("MEMO_CACHE", "1").lower()
    return {"0": "off", "off": "off", "no": "off", "read": "read", "sync": "sync"}.get(v, "on")


def cache_path(in_file, record):
    safe = "".join(ch if (ch.isalnum() or ch in "._-") else "_%02x" % ord(ch) for ch in record) or "_"
    return os.path.join(in_file + ".memo", safe + ".v%d.pk" % VERSION)


def _source_key(in_file):
    st = os.stat(in_file)
    return {"size": st.st_size, "mtime_ns": st.st_mtime_ns}


def _align(x, a=4096):
    return (x + a - 1) // a * a


def write(in_file, record, ix):
    """ix: a packed DeviceIndex holding EVERY row of `record`.  Writes the cache file atomically."""
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
    off_pk = HEADER_BYTES
    off_pa = _align(off_pk + pk.nbytes)
    off_boff = _align(off_pa + pa.nbytes)
    off_long = _align(off_boff + boff.nbytes)
    total = off_long + longs.nbytes
    head = {"version": VERSION, "record": record, "source": _source_key(in_file), "rows": rows, "format": fmt,
            "bucket_shift": inf["bucket_shift"], "buckets": nb, "min_start": inf["min_start"], "max_start": inf["max_start"],
            "max_annot": inf["max_annot"], "long_rows": n_long, "off_pk": off_pk, "off_pa": off_pa, "off_boff": off_boff,
            "off_long": off_long, "bytes": total}
    blob = MAGIC + json.dumps(head).encode()
    if len(blob) > HEADER_BYTES:
        raise ValueError("cache header too large")
    path = cache_path(in_file, record)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "wb") as fh:
        fh.write(blob.ljust(HEADER_BYTES, b"\0"))
        for off, arr in ((off_pk, pk), (off_pa, pa), (off_boff, boff), (off_long, longs)):
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
        if os.path.getsize(path) != head["bytes"]:
            return None
        return head, np.memmap(path, dtype=np.uint8, mode="r")
    except (OSError, ValueError, KeyError):
        return None


def load_region(in_file, record, query_start, query_end, device=0):
    """DeviceIndex with the rows of `record` that have query_start < start < query_end (and a few more from
    the two buckets at the edges, which the sweep ignores as it ignores every row outside the window), from
    the cache; None when there is no valid cache."""
    got = _open(in_file, record)
    if got is None:
        return None
    head, mm = got
    rows, nb, shift = head["rows"], head["buckets"], head["bucket_shift"]
    boff = mm[head["off_boff"]:head["off_boff"] + 8 * nb].view(np.int64)
    b_lo = min(max(query_start, 0) >> shift, nb - 1)
    b_hi = min(max((max(query_end, 0) >> shift) + 1, b_lo), nb - 1)
    r0, r1 = int(boff[b_lo]), int(boff[b_hi])
    table = boff[b_lo:b_hi + 1]                      # absolute entries: the library rebases them to r0
    pk = mm[head["off_pk"] + 4 * r0:head["off_pk"] + 4 * r1]
    pa = mm[head["off_pa"] + 2 * r0:head["off_pa"] + 2 * r1] if head["format"] == 6 else None
    n_long = head["long_rows"]
    longs = np.array(mm[head["off_long"]:head["off_long"] + 24 * n_long].view(np.int64)) if n_long else None
    h = C.c_void_p()
    n = r1 - r0
    lo, hi = slice_extent(head, b_lo, b_hi, n)
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


def build(in_file, record, device=0):
    """decode every row of `record` from the Parquet file, pack it with the library, write the cache"""
    from .memo_query import region_index
    hi = (1 << 61) - 1
    ix = region_index(in_file, record, -1, hi, device=device, k=2, use_cache=False)
    with ix:
        inf = ix.info()
        if inf["packed_format"] not in (4, 6, 12) or inf["rows"] == 0:
            return None                     # unpackable rows (or none): nothing worth caching
        return write(in_file, record, ix)


def build_in_background(in_file, record):
    import subprocess
    env = dict(os.environ, MEMO_CACHE="read")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    try:
        subprocess.Popen([sys.executable, "-m", "memo_amd.cache", "build", in_file, record], env=env,
                         stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                         start_new_session=True)
    except OSError:
        pass


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "build":
        dev = int(os.environ.get("MEMO_DEVICE", "0"))
        for rec in sys.argv[3:]:
            print(build(sys.argv[2], rec, dev))
    else:
        sys.exit("usage: python -m memo_amd.cache build INDEX.parquet RECORD [RECORD ...]")
