"""Access to the committed golden fixtures (tests/golden/, made by tools/make_golden.py)."""
import functools
import hashlib
import json
import os

import numpy as np
import pyarrow.parquet as pq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(None)
def manifest():
    return json.load(open(os.path.join(GOLD, "manifest.json")))


def cases(membership=None, raises=None):
    out = []
    for c in manifest():
        if membership is not None and c["membership"] != membership:
            continue
        if raises is not None and ("raises" in c) != raises:
            continue
        out.append(c)
    return out


def region(c):
    rec, se = c["region"].split(":")
    qs, qe = map(int, se.split("-"))
    return rec, qs, qe


@functools.lru_cache(None)
def index_columns(index, record):
    """All rows of one chromosome of a golden index, in file order: (start, end, order) int64."""
    t = pq.read_table(os.path.join(GOLD, index))
    chrom = np.asarray(t.column("f0").to_pylist(), dtype=object)
    sel = chrom == record
    return tuple(np.ascontiguousarray(t.column(c).to_numpy()[sel], np.int64) for c in ("f1", "f2", "f3"))


def load(c):
    z = np.load(os.path.join(GOLD, c["npz"]))
    return {k: z[k] for k in z.files}


def expected_matrix(c, z):
    """membership golden -> uint8 [L, N]."""
    return np.unpackbits(z["bits"], axis=1)[:, :c["n"]]


def out_bytes(c):
    return open(os.path.join(GOLD, c["out"]), "rb").read() if "out" in c else None


def sha(b):
    return hashlib.sha256(b).hexdigest()
