"""`memo view` preprocessing on the GPU: the per-bin composition of a conservation vector.

Counterpart of /root/reference/src/plot_conservation.py:46-65 (preprocess_data).  The histogram
(the only part that touches every position) runs on the device straight from a conservation
result that is still in HBM, or from a vector read from the reference's text file; normalisation
and the long-format table are host-side and follow the reference line by line:

  bin edges   list(map(int, np.linspace(0, positions, n_bins + 1)))            :52
  per bin     count[order] / (positions in the bin)  for order in 0..n_docs     :55-56
              (an empty bin raises ZeroDivisionError, as in the reference)
  table       melt over orders, bins innermost; rows of order == n_docs dropped :60-65
Plotting itself (plotnine) is not part of this build.
"""
import ctypes as C

import numpy as np

from ._lib import check, lib


def bin_edges(positions, n_bins):
    return np.array(list(map(int, np.linspace(0, positions, n_bins + 1))), np.int64)


def bin_counts(vec, n_docs, n_bins, device=0, stream=None):
    """uint64 [n_bins, n_docs + 1] histogram.  vec: host uint16 array, or (device pointer, length)."""
    if isinstance(vec, tuple):
        d_vec, L, tmp = C.c_void_p(int(vec[0])), int(vec[1]), None
    else:
        host = np.ascontiguousarray(vec, np.uint16)
        L, tmp = len(host), C.c_void_p()
        check(lib().memo_dev_malloc(device, host.nbytes, C.byref(tmp)))
        d_vec = tmp
        check(lib().memo_dev_upload(device, d_vec, host.ctypes.data, host.nbytes, None))
    try:
        edges = bin_edges(L, n_bins)
        counts = np.zeros((n_bins, n_docs + 1), np.uint64)
        check(lib().memo_bin_conservation_dev(d_vec, L, edges.ctypes.data, n_bins, n_docs, counts.ctypes.data,
                                              device, None if stream is None else C.c_void_p(int(stream))))
    finally:
        if tmp is not None:
            lib().memo_dev_free(device, tmp)
    return counts, edges


def read_conservation_text(path):
    """the reference's out.txt: one integer per line"""
    return np.loadtxt(path, dtype=np.int64, ndmin=1).astype(np.uint16)


def preprocess_data(vec_or_path, n_docs, n_bins, device=0):
    """Same table as the reference's preprocess_data: dict of arrays 'bin' (int64),
    'No. Genomes' (float64), 'value' (float64), rows ordered as pd.melt leaves them."""
    vec = read_conservation_text(vec_or_path) if isinstance(vec_or_path, str) else vec_or_path
    counts, edges = bin_counts(vec, n_docs, n_bins, device)
    width = np.diff(edges)
    if np.any(width == 0):
        raise ZeroDivisionError("division by zero")        # Counter of an empty bin (:56)
    value = counts.astype(np.float64) / width[:, None].astype(np.float64)      # [bin, order]
    orders = np.arange(n_docs)                               # order == n_docs is dropped (:65)
    return {"bin": np.tile(np.arange(n_bins, dtype=np.int64), n_docs),
            "No. Genomes": np.repeat(orders.astype(np.float64), n_bins),
            "value": value[:, :n_docs].T.reshape(-1)}
