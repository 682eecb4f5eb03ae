"""ctypes binding of the C ABI declared in include/memo_amd.h, memo_amd_multi.h, memo_amd_dap.h and
memo_amd_transport.h.

There is NO fallback: if libmemo_amd.so is missing or a HIP call fails, this raises.
The CPU restatement under oracle/ is test infrastructure and is never imported here.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MEMO_AMD_LIB") or os.path.join(_HERE, "libmemo_amd.so")   # override: A/B of builds
# the product objects + the A/B switches of include/memo_amd_debug.h (tests, fuzzers, tools/ab.py)
AB_SO_PATH = os.environ.get("MEMO_AMD_AB_LIB") or os.path.join(_HERE, "libmemo_amd_ab.so")

MEMO_OK, MEMO_EINVAL, MEMO_EHIP, MEMO_ENOTREADY, MEMO_EUNSORTED, MEMO_ELONGROW, MEMO_EUNPACKABLE = 0, -1, -2, -3, -4, -5, -6


class MemoError(RuntimeError):
    """A C-ABI call returned a negative code (message from memo_last_error())."""

    def __init__(self, code, msg):
        super().__init__(f"memo_amd error {code}: {msg}")
        self.code = code


class MemoIndexError(MemoError, IndexError):
    """The reference raises IndexError here (annot column outside the result matrix)."""


class MemoValueError(MemoError, ValueError):
    """The reference raises ValueError here (window end before window start)."""


class MemoUnpackable(MemoError):
    """memo_builder_*: the rows cannot take the packed way in; upload them as int64 columns."""


class IndexInfo(C.Structure):
    """memo_index_info_t (include/memo_amd.h): versioned -- set struct_bytes before memo_index_get_info_v5"""
    _fields_ = [("struct_bytes", C.c_uint32), ("version", C.c_uint32),
                ("rows", C.c_uint64), ("min_start", C.c_int64), ("max_start", C.c_int64),
                ("device", C.c_int32), ("bucket_shift", C.c_int32), ("buckets", C.c_uint64),
                ("was_sorted", C.c_int32), ("finalized", C.c_int32), ("device_bytes", C.c_uint64),
                ("packed_format", C.c_int32), ("has_wide", C.c_int32), ("pack_ms", C.c_float),
                ("dense_rows", C.c_int32), ("long_rows", C.c_uint64), ("max_annot", C.c_uint64),
                ("bucket_base", C.c_int64), ("last_sweep", C.c_int32), ("last_variant", C.c_int32), ("dense_row_count", C.c_uint64), ("last_rows_read", C.c_uint64),
                ("last_view_ms", C.c_float), ("row_order", C.c_int32), ("side_bytes", C.c_uint64),
                ("views_resident", C.c_int32), ("tile_tables_resident", C.c_int32), ("view_builds", C.c_uint64),
                ("last_level_arrays", C.c_int32), ("last_view_placed", C.c_int32), ("view_placings", C.c_uint64),
                ("last_view_rows_per_group", C.c_int32), ("reserved", C.c_int32)]


# every symbol the product headers declare: name -> (restype, argtypes)
_P, _I32, _I64, _U64, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t
SYMBOLS = {
    "memo_last_error": (C.c_char_p, []),
    "memo_device_count": (C.c_int, []),
    "memo_version": (C.c_char_p, []),
    "memo_host_threads": (C.c_int, [C.POINTER(_I32), C.POINTER(C.c_double)]),
    "memo_index_create": (C.c_int, [_U64, _I32, C.POINTER(_P)]),
    "memo_index_upload": (C.c_int, [_P, _P, _P, _P, _U64]),
    "memo_index_upload_rows": (C.c_int, [_P, _U64, _P, _P, _P, _U64]),
    "memo_index_truncate": (C.c_int, [_P, _U64]),
    "memo_index_columns": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "memo_index_finalize": (C.c_int, [_P, _I32, _I32]),
    "memo_index_pack": (C.c_int, [_P, _I32]),
    "memo_index_pack_dense": (C.c_int, [_P, _I32]),
    "memo_index_get_info_v5": (C.c_int, [_P, C.POINTER(IndexInfo)]),
    "memo_index_set_option": (C.c_int, [_P, _I32, _I64]),
    "memo_index_prepare": (C.c_int, [_P, _I32, _I32, _I32, _I64, _P, C.POINTER(_U64)]),
    "memo_index_export_packed": (C.c_int, [_P, _P, _P, _P, _P]),
    "memo_index_import_packed": (C.c_int, [_U64, _I32, _I32, _I64, _P, _P, _P, _U64, _I64, _I64, _I64, _U64, _P, _U64,
                                           C.POINTER(_P)]),
    "memo_index_export_dense": (C.c_int, [_P, _P, _P, _P]),
    "memo_index_export_view": (C.c_int, [_P, _I32, _I32, _P, _P, _P, _P, _P]),
    "memo_index_import_dense": (C.c_int, [_U64, _I32, _I32, _I64, _P, _P, _U64, _I64, _I64, _I64, _U64, _P, _U64,
                                          C.POINTER(_P)]),
    "memo_dense_rows_can_answer": (C.c_int, [_U64, _I64, _I64, _U64, _I32, _I32, _I32]),
    "memo_index_destroy": (None, [_P]),
    "memo_builder_create": (C.c_int, [_U64, _I32, _I32, C.POINTER(_P)]),
    "memo_builder_create_rows": (C.c_int, [_U64, _I32, _I32, _I32, C.POINTER(_P)]),
    "memo_builder_push": (C.c_int, [_P, _P, _P, _P, _U64]),
    "memo_builder_push_rows": (C.c_int, [_P, _P, _U64]),
    "memo_builder_finish": (C.c_int, [_P, C.POINTER(_P)]),
    "memo_builder_destroy": (None, [_P]),
    "memo_query_conservation_dev": (C.c_int, [_P, _I64, _I64, _I32, _I32, _P, _P]),
    "memo_query_membership_dev": (C.c_int, [_P, _I64, _I64, _I32, _I32, _P, _P]),
    "memo_query_conservation_u8_dev": (C.c_int, [_P, _I64, _I64, _I32, _I32, _P, _P]),
    "memo_query_check": (C.c_int, [_P, _P]),
    "memo_conservation": (C.c_int, [_P, _P, _P, _U64, _I64, _I64, _I32, _I32, _P, _I32]),
    "memo_membership": (C.c_int, [_P, _P, _P, _U64, _I64, _I64, _I32, _I32, _P, _I32]),
    "memo_conservation_rows": (C.c_int, [_P, _U64, _I64, _I64, _I32, _I32, _P, _I32]),
    "memo_membership_rows": (C.c_int, [_P, _U64, _I64, _I64, _I32, _I32, _P, _I32]),
    "memo_split_window": (C.c_int, [_I64, _I64, _I32, _I32, C.c_double, _P]),
    "memo_conservation_multi": (C.c_int, [_P, _P, _P, _U64, _I64, _I64, _I32, _I32, _P, _P, _I32]),
    "memo_membership_multi": (C.c_int, [_P, _P, _P, _U64, _I64, _I64, _I32, _I32, _P, _P, _I32]),
    "memo_query_conservation_multi_dev": (C.c_int, [_P, _I32, _I64, _I64, _I32, _I32, _P, _I32, _P, C.c_double]),
    "memo_query_membership_multi_dev": (C.c_int, [_P, _I32, _I64, _I64, _I32, _I32, _P, _I32, _P, C.c_double]),
    "memo_dev_malloc": (C.c_int, [_I32, _SZ, C.POINTER(_P)]),
    "memo_dev_free": (C.c_int, [_I32, _P]),
    "memo_dev_upload": (C.c_int, [_I32, _P, _P, _SZ, _P]),
    "memo_dev_download": (C.c_int, [_I32, _P, _P, _SZ, _P]),
    "memo_emit_conservation": (_SZ, [_P, _I64, _P, _SZ]),
    "memo_emit_membership": (_SZ, [_P, _I64, _I32, _P, _SZ]),
    "memo_bin_conservation_dev": (C.c_int, [_P, _I64, _P, _I32, _I32, _P, _I32, _P]),
    "memo_dap_create": (C.c_int, [_I32, _P, _I32, _I32, _I32, _I32, C.POINTER(_P)]),
    "memo_dap_push": (C.c_int, [_P, _P, _I64, C.POINTER(_U64)]),
    "memo_dap_fetch": (C.c_int, [_P, _P, _P, _P, _P]),
    "memo_dap_finish": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(_U64)]),
    "memo_dap_destroy": (None, [_P]),
    "memo_parse_ints": (C.c_int64, [_P, _SZ, _P, _SZ]),
    "memo_emit_bed": (_SZ, [_P, _P, _P, _P, _U64, _P, _I32, _P, _SZ]),
    "memo_transport_runs_bytes": (_SZ, [_I64, C.c_uint32]),
    "memo_transport_runs_pack_dev": (C.c_int, [_P, _I64, C.c_uint32, _P, _I32, _P]),
    "memo_transport_runs_unpack_dev": (C.c_int, [_P, _I64, C.c_uint32, _P, _I32, _P]),
    "memo_transport_runs_stats": (C.c_int, [_P, _I32, _P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "memo_transport_runs16_pack_dev": (C.c_int, [_P, _I64, C.c_uint32, _P, _I32, _P]),
    "memo_transport_runs16_unpack_dev": (C.c_int, [_P, _I64, C.c_uint32, _P, _I32, _P]),
    "memo_transport_runs_unpack_many_dev": (C.c_int, [_P, _P, _I32, _I64, C.c_uint32, _I32, _I32, _P]),
    "memo_transport_bytes": (_SZ, [_I64, C.c_uint32]),
    "memo_transport_pack_dev": (C.c_int, [_P, _I64, C.c_uint32, _P, _I32, _P]),
    "memo_transport_unpack_dev": (C.c_int, [_P, _I64, _P, _I32, _P]),
    "memo_transport_exceptions": (C.c_int, [_P, _I32, _P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "memo_transport_dense_bytes": (_SZ, [_I64, C.c_uint32, C.c_uint32]),
    "memo_transport_dense_pack_dev": (C.c_int, [_P, _I64, C.c_uint32, C.c_uint32, _P, _I32, _P]),
    "memo_transport_dense_unpack_dev": (C.c_int, [_P, _I64, C.c_uint32, C.c_uint32, _P, _I32, _P]),
    "memo_transport_dense_stats": (C.c_int, [_P, _I32, _P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "memo_synth_fill": (C.c_int, [_P, _U64, _U64, _U64, _I32, _U64]),
}

# include/memo_amd_debug.h: exported by libmemo_amd_ab.so only
DEBUG_SYMBOLS = {
    "memo_debug_set_tuning": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32]),
    "memo_debug_stream_rows": (C.c_int, [_P, _P]),
    "memo_debug_row_order": (C.c_int, [_P, _I32]),
    "memo_debug_no_views": (C.c_int, [_P, _I32]),
    "memo_debug_view_colouring": (C.c_int, [_I32]),
    "memo_debug_six_views": (C.c_int, [_I32]),
    "memo_debug_fail_side_allocations": (C.c_int, [_I32]),
    "memo_debug_dense_keep_all": (C.c_int, [_I32]),
    "memo_debug_one_shot_way": (C.c_int, [_I32]),
    "memo_debug_last_one_shot_sweep": (C.c_int, []),
    "memo_debug_set_stamp_buffer": (C.c_int, [_P]),
}


def build(force=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "-s", "clean"])
    subprocess.check_call(["make", "-C", src, "-s", "-j", str(min(4, os.cpu_count() or 1))])
    return SO_PATH


_lib = None
_product = None
_ab = None


def _bind(path, symbols):
    L = C.CDLL(path)
    for name, (res, args) in symbols.items():
        f = getattr(L, name)        # AttributeError if the library does not export it
        f.restype = res
        f.argtypes = args
    return L


def use_ab(on=True):
    """Make lib() hand out libmemo_amd_ab.so (product objects + the A/B switches) instead of the
    product library -- for tests and tools that set kernel shapes.  Returns the library now in use."""
    global _lib, _ab
    lib()
    if on:
        if _ab is None:
            if not os.path.exists(AB_SO_PATH):
                build()
            _ab = _bind(AB_SO_PATH, {**SYMBOLS, **DEBUG_SYMBOLS})
        _lib = _ab
    else:
        _lib = _product
    return _lib


def lib():
    global _lib, _product
    if _lib is None:
        if not os.path.exists(SO_PATH) and not os.environ.get("MEMO_AMD_LIB"):
            try:                                  # compile the real thing; never substitute for it
                build()
            except Exception:
                pass
        if not os.path.exists(SO_PATH):
            raise ImportError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                              "g.build()'` or `make -C memo_amd/csrc` -- there is no CPU fallback")
        _product = _lib = _bind(SO_PATH, SYMBOLS)
    return _lib


def check(rc):
    if rc < 0:
        msg = lib().memo_last_error().decode(errors="replace")
        if rc == MEMO_EINVAL and "IndexError" in msg:
            raise MemoIndexError(rc, msg)
        if rc == MEMO_EINVAL and "ValueError" in msg:
            raise MemoValueError(rc, msg)
        if rc == MEMO_EUNPACKABLE:
            raise MemoUnpackable(rc, msg)
        raise MemoError(rc, msg)
    return rc
