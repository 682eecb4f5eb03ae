"""Host-side native code under AddressSanitizer + UBSan (CPU build only: GPU sanitizers are not
available on this pool).  Covers the C oracle and the product's host C++ (text emitters, DAP
parser), each built on its own with gcc -fsanitize and driven from a child process that preloads
libasan."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="libasan not installed")
def test_host_cpp_under_asan(tmp_path):
    so = str(tmp_path / "libmemo_emit_asan.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-pthread", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"), "-shared", "-o", so,
                           os.path.join(ROOT, "memo_amd", "csrc", "memo_emit.cpp")])
    script = textwrap.dedent(f"""
        import ctypes as C, numpy as np, sys
        sys.path.insert(0, {ROOT!r})
        from oracle import memo_oracle as O
        L = C.CDLL({so!r})
        L.memo_emit_conservation.restype = C.c_size_t
        L.memo_emit_membership.restype = C.c_size_t
        L.memo_emit_bed.restype = C.c_size_t
        L.memo_parse_ints.restype = C.c_int64
        P = C.c_void_p
        rng = np.random.default_rng(3)
        for n in (0, 1, 7, 100000, 3_000_000):
            v = rng.integers(0, 65535, n).astype(np.uint16)
            need = L.memo_emit_conservation(P(v.ctypes.data), C.c_int64(n), None, C.c_size_t(0))
            buf = np.empty(need, np.uint8)
            L.memo_emit_conservation(P(v.ctypes.data), C.c_int64(n), P(buf.ctypes.data), C.c_size_t(need))
            assert buf.tobytes() == O.emit_conservation(v), n
        for n, nd in ((0, 5), (3, 1), (1000, 33), (20000, 100), (5, 500)):
            b = rng.integers(0, 2**32, (n, (nd + 31) // 32), dtype=np.uint64).astype(np.uint32)
            need = L.memo_emit_membership(P(b.ctypes.data), C.c_int64(n), C.c_int32(nd), None, C.c_size_t(0))
            buf = np.empty(max(need, 1), np.uint8)
            L.memo_emit_membership(P(b.ctypes.data), C.c_int64(n), C.c_int32(nd), P(buf.ctypes.data), C.c_size_t(need))
            assert buf[:need].tobytes() == O.emit_membership(b, nd), (n, nd)
        text = (" ".join(map(str, rng.integers(-5, 10**9, 200000))) + "\\n").encode() * 3
        cnt = L.memo_parse_ints(text, C.c_size_t(len(text)), None, C.c_size_t(0))
        out = np.empty(cnt, np.int64)
        L.memo_parse_ints(text, C.c_size_t(len(text)), P(out.ctypes.data), C.c_size_t(cnt))
        assert np.array_equal(out, np.array(text.split(), np.int64))
        assert L.memo_parse_ints(b"1 2 x", C.c_size_t(5), None, C.c_size_t(0)) == -1
        n = 50000
        rec = rng.integers(0, 3, n).astype(np.int32); st = rng.integers(0, 10**9, n); en = st + rng.integers(0, 99, n)
        an = rng.integers(1, 500, n).astype(np.int32); names = ["chr1", "a_long_record_name", "x"]
        blob = b"".join(s.encode() + b"\\0" for s in names)
        need = L.memo_emit_bed(P(rec.ctypes.data), P(st.ctypes.data), P(en.ctypes.data), P(an.ctypes.data), C.c_uint64(n), blob, C.c_int32(3), None, C.c_size_t(0))
        buf = np.empty(need, np.uint8)
        L.memo_emit_bed(P(rec.ctypes.data), P(st.ctypes.data), P(en.ctypes.data), P(an.ctypes.data), C.c_uint64(n), blob, C.c_int32(3), P(buf.ctypes.data), C.c_size_t(need))
        want = "".join(f"{{names[r]}}\\t{{s}}\\t{{e}}\\t{{a}}\\n" for r, s, e, a in zip(rec, st, en, an)).encode()
        assert buf.tobytes() == want
        print("asan ok")
    """)
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0", MEMO_EMIT_THREADS="4")
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "asan ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.skipif(_libasan() is None, reason="libasan not installed")
def test_oracle_under_asan(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0",
               MEMO_ORACLE_LIB=os.path.join(ROOT, "oracle", "libmemo_oracle_asan.so"))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-q", "-x",
                        "-k", "ex_ or rnd_n4 or negoverlap or synth or split"], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_host_packer_under_sanitizers(tmp_path, sanitizer):
    """memo_amd/csrc/memo_hostcore.cpp -- worker pool, pinned ring, row packers (4-byte words and dense rows), the
    builder's push loop: the library's only multi-threaded host code -- with the device seam stubbed by
    tests/host_stub.cpp (memcpy on copier threads), under ASan + UBSan and under ThreadSanitizer: two builders on two
    threads, ragged pieces, the late switch to 12-bit annots, refusals, the pipelined transfers, and one push of many chunks
    (the push loop as ONE job of the pool: workers packing ahead into four slots, the caller issuing the copies)."""
    lib = subprocess.run(["gcc", "-print-file-name=lib%s.so" % ("tsan" if sanitizer == "thread" else "asan")],
                         capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(lib) and os.path.exists(lib)):
        pytest.skip("sanitizer runtime not installed")
    exe = str(tmp_path / "hostcore")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-DMEMO_HOST_TEST_KNOBS", "-fsanitize=" + sanitizer,
                           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "host_stub.cpp"),
                           os.path.join(ROOT, "memo_amd", "csrc", "memo_hostcore.cpp"), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=1", MEMO_HOST_THREADS="6")
    for simd in ("2", "1", "0"):     # the row pass: the hand-written AVX-512 packers, the compiler's AVX2 instance (where the CPU has them), the plain one
        r = subprocess.run([exe], capture_output=True, text=True, env=dict(env, MEMO_HOST_SIMD=simd), timeout=600)
        assert r.returncode == 0 and "hostcore ok" in r.stdout, (simd, (r.stdout + r.stderr)[-3000:])
        assert "WARNING: ThreadSanitizer" not in r.stderr and "runtime error" not in r.stderr, (simd, r.stderr[-3000:])
