"""memo_amd -- MI355X-native implementation of MEMO's windowed k-mer query path.

Layout (only what the path needs):
  csrc/            HIP kernels for gfx950 + the C ABI (include/memo_amd*.h)
  _lib.py          ctypes binding of libmemo_amd.so -- raises if the library is missing
  index.py         DeviceIndex: an index chromosome resident in HBM; IndexBuilder; sweep launches
  memo_query.py    host-side mirror of the reference's src/memo_query.py (same function
                   names and argument meaning: filter_pq, memo_init, memo_query, print_res)
  cache.py         sidecar cache of a record's packed rows next to the Parquet index
  _fastquery.py    `memo query` answered from that cache with nothing but ctypes (no NumPy import)
  synth.py         the synthetic pangenome workloads of BASELINE.json
  shard.py         window sharding across ranks + gather (torch.distributed)

Attributes are loaded on first use (PEP 562), so that `import memo_amd._fastquery` -- the CLI's cache-hit
path -- does not pay for NumPy.
"""
__version__ = "0.2.0"

_LAZY = {
    "MemoError": "_lib", "MemoUnpackable": "_lib", "build": "_lib", "lib": "_lib",
    "DeviceIndex": "index", "IndexBuilder": "index", "conservation": "index", "membership": "index",
    "conservation_rows": "index", "membership_rows": "index",
    "emit_conservation": "index", "emit_membership": "index",
}


def __getattr__(name):
    import importlib
    if name in _LAZY:
        value = getattr(importlib.import_module("." + _LAZY[name], __name__), name)
        globals()[name] = value
        return value
    if name in ("_lib", "index", "memo_query", "cache", "synth", "shard", "view", "dap_to_bed", "_fastquery"):
        return importlib.import_module("." + name, __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return sorted(list(globals()) + list(_LAZY))
