#!/usr/bin/env python3
"""DAP -> MEM / MEM-overlap BED rows on the GPU; same command line and stdout as the reference's
src/dap_to_bed.py for the forms `memo index` uses:

    dap_to_bed.py --mem [--order] [--overlap] --fai pivot.fa.fai --dap dap.txt > index.bed

(/root/reference/src/dap_to_bed.py:140-185; --ms is refused: that branch of the reference calls an
undefined name and cannot run.)  The DAP text is read in blocks, parsed to an int32 matrix and
pushed through memo_dap_* (memo_amd/csrc/memo_dap.hip); rows come back in the reference's print
order and are formatted by memo_emit_bed.  `dap_to_parquet` writes the Parquet index directly
(schema of parquet_compress_bed.py:19-38) without the BED detour.

Limits the reference does not have (each raises, none wraps silently): a record shorter than 2^30
positions, at most 4096 DAP columns, DAP rows numbered 0, 1, 2, ... in the first column, values in
[0, 2^31).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

from ._lib import check, lib


def parse_fai(fai_path):
    """(names, cumulative record offsets) -- dap_to_bed.py:20-28"""
    names, lens = [], []
    with open(fai_path) as fh:
        for row in fh:
            row = row.strip()
            if row:
                f = row.split()
                names.append(f[0])
                lens.append(int(f[1]))
    return names, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def parse_ints(text):
    """all whitespace-separated integers of a bytes block (memo_parse_ints: C++, multi-threaded)"""
    n = lib().memo_parse_ints(text, len(text), None, 0)
    if n < 0:
        raise ValueError("DAP text holds something that is not an integer")
    out = np.empty(n, np.int64)
    lib().memo_parse_ints(text, len(text), out.ctypes.data, n)
    return out


def dap_blocks(dap_path, block_bytes=256 << 20):
    """int64 [rows, 1 + columns] blocks of the space-separated DAP text (index.sh:83)"""
    with open(dap_path, "rb") as fh:
        first = fh.readline()
        if not first.strip():
            return
        width = len(first.split())
        fh.seek(0)
        tail = b""
        while True:
            chunk = fh.read(block_bytes)
            if not chunk:
                break
            chunk = tail + chunk
            cut = chunk.rfind(b"\n") + 1          # whole lines only; the rest waits for the next read
            tail, chunk = chunk[cut:], chunk[:cut]
            flat = parse_ints(chunk)
            if flat.size % width:
                raise ValueError("DAP rows differ in their number of columns")
            if flat.size:
                yield flat.reshape(-1, width)
        if tail.strip():
            flat = parse_ints(tail)
            if flat.size % width:
                raise ValueError("DAP rows differ in their number of columns")
            yield flat.reshape(-1, width)


class DapConverter:
    def __init__(self, columns, rec_begin, order, overlap, device=0):
        self._h = C.c_void_p()
        self.columns = columns
        rb = np.ascontiguousarray(rec_begin, np.int64)
        check(lib().memo_dap_create(columns, rb.ctypes.data, len(rb) - 1, int(order), int(overlap), device, C.byref(self._h)))

    @staticmethod
    def _bufs(n):
        return (np.empty(n, np.int32), np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int32))

    def push(self, lcp):
        lcp = np.asarray(lcp)
        if lcp.size and lcp.dtype != np.int32:
            # matching statistics are lengths inside a record (< 2^30 here): anything else would wrap
            # silently in the int32 matrix the device works on
            lo, hi = int(lcp.min()), int(lcp.max())
            if lo < 0 or hi >= 2 ** 31:
                raise ValueError(f"DAP values must lie in [0, 2^31): found {lo if lo < 0 else hi}")
        lcp = np.ascontiguousarray(lcp, np.int32)
        n = C.c_uint64()
        check(lib().memo_dap_push(self._h, lcp.ctypes.data, lcp.shape[0], C.byref(n)))
        out = self._bufs(n.value)
        if n.value:
            check(lib().memo_dap_fetch(self._h, *[a.ctypes.data for a in out]))
        return out

    def finish(self):
        out = self._bufs(self.columns)
        n = C.c_uint64()
        check(lib().memo_dap_finish(self._h, *[a.ctypes.data for a in out], C.byref(n)))
        return tuple(a[:n.value] for a in out)

    def close(self):
        if self._h:
            lib().memo_dap_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def bed_bytes(names, rec, start, end, annot):
    blob = b"".join(n.encode() + b"\0" for n in names)
    args = (rec.ctypes.data, start.ctypes.data, end.ctypes.data, annot.ctypes.data, len(rec), blob, len(names))
    need = lib().memo_emit_bed(*args, None, 0)
    buf = np.empty(need, np.uint8)
    lib().memo_emit_bed(*args, buf.ctypes.data, need)
    return buf


def convert(dap_path, fai_path, order, overlap, device=0):
    """yields (rec, start, end, annot) row batches in print order"""
    names, rec_begin = parse_fai(fai_path)
    conv, nxt = None, 0
    try:
        for block in dap_blocks(dap_path):
            if conv is None:
                conv = DapConverter(block.shape[1] - 1, rec_begin, order, overlap, device)
            if block[0, 0] != nxt or np.any(np.diff(block[:, 0]) != 1):
                raise ValueError("DAP rows must be numbered 0, 1, 2, ... (nl -v0, index.sh:83)")
            if block[-1, 0] >= rec_begin[-1]:       # dap_to_bed.py:72-73
                raise Exception("Position beyond all intervals; ensure your fai file is from fasta of initial query.")
            nxt = int(block[-1, 0]) + 1
            yield names, conv.push(block[:, 1:])
        if conv is not None:
            yield names, conv.finish()
    finally:
        if conv is not None:
            conv.close()


def dap_to_parquet(dap_path, fai_path, out_path, order, device=0, codec="ZSTD"):
    """the index file `memo index` ends with: columns f0 utf8, f1 f2 f3 int64, ZSTD"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    schema = pa.schema([("f0", pa.utf8()), ("f1", pa.int64()), ("f2", pa.int64()), ("f3", pa.int64())])
    with pq.ParquetWriter(out_path, schema, compression=codec) as w:
        for names, (rec, start, end, annot) in convert(dap_path, fai_path, order, True, device):
            if len(rec):
                f0 = pa.DictionaryArray.from_arrays(pa.array(rec, pa.int32()), pa.array(names, pa.utf8())).cast(pa.utf8())
                w.write_table(pa.table({"f0": f0, "f1": start, "f2": end, "f3": annot.astype(np.int64)}, schema=schema))


################################################################################

def parse_arguments(argv=None):
    parser = argparse.ArgumentParser(description="Takes in .fai and full document array profile and converts to bed-style MEM intervals to stdout.")
    parser.add_argument('--fai', dest='fai_path', help='path to fai file', required=True)
    parser.add_argument('--dap', dest='dap_path', help='path to full document profile', required=True)
    parser.add_argument("--ms", action="store_true", default=False, dest="print_ms", help="Extract matching statistics and print to stdout (it can either MSs or MEMs, not both).")
    parser.add_argument("--mem", action="store_true", default=False, dest="print_mems", help="Extract MEMs and print to stdout (it can either MSs or MEMs, not both).")
    parser.add_argument("--overlap", action="store_true", default=False, dest="print_overlaps", help="extract overlap MEMs (can only be used when extracting MEMs).")
    parser.add_argument("--order", action="store_true", default=False, dest="sort_lcps", help="sort LCP row to extract order MS/MEMs.")
    return parser.parse_args(argv)


def check_args(args):
    if not os.path.isfile(args.fai_path):
        raise Exception("The fai file does not exist.")
    if not os.path.isfile(args.dap_path):
        raise Exception("The dap file does not exist.")
    if not args.fai_path.endswith(".fai"):
        raise Exception("The fai file has the incorrect file extension.")
    if (args.print_ms + args.print_mems) != 1:
        raise Exception("Error: Either print MSs or MEMs, not both.")
    if args.print_overlaps and args.print_ms:
        raise Exception("Error: Can only print overlaps if printing MEMs.")
    if args.print_ms:
        raise Exception("--ms is not supported: the reference's own --ms branch cannot run (dap_to_bed.py:51)")


def main(args, out=None):
    out = out or sys.stdout.buffer
    for names, rows in convert(args.dap_path, args.fai_path, args.sort_lcps, args.print_overlaps,
                               int(os.environ.get("MEMO_DEVICE", "0"))):
        if len(rows[0]):
            out.write(memoryview(bed_bytes(names, *rows)))
    out.flush()


if __name__ == "__main__":
    a = parse_arguments()
    check_args(a)
    main(a)
