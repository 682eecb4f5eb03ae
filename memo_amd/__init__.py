"""memo_amd -- MI355X-native implementation of MEMO's windowed k-mer query path.

Layout (only what the path needs):
  csrc/            HIP kernels for gfx950 + the C ABI (include/memo_amd.h)
  _lib.py          ctypes binding of libmemo_amd.so -- raises if the library is missing
  index.py         DeviceIndex: an index chromosome resident in HBM; sweep launches
  memo_query.py    host-side mirror of the reference's src/memo_query.py (same function
                   names and argument meaning: filter_pq, memo_init, memo_query, print_res)
  synth.py         the synthetic pangenome workloads of BASELINE.json
  shard.py         window sharding across ranks + gather (torch.distributed)
"""
from ._lib import MemoError, MemoUnpackable, build, lib  # noqa: F401
from .index import DeviceIndex, IndexBuilder, conservation, membership, emit_conservation, emit_membership  # noqa: F401

__version__ = "0.1.0"
