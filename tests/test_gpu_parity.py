"""Parity of the HIP path with the oracle and with the committed golden vectors.
Everything here calls through the C ABI (memo_amd._lib) and needs a real MI355X."""
import os

import numpy as np
import pytest

from tests import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def memo():
    import memo_amd
    from memo_amd import _lib
    memo_amd.build()                     # make: a no-op when libmemo_amd.so is up to date
    assert _lib.lib().memo_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return memo_amd


@pytest.fixture
def ab(memo):
    """tests that set kernel shapes run against libmemo_amd_ab.so (product objects + memo_debug.o)"""
    from memo_amd import _lib
    _lib.use_ab(True)
    yield _lib
    _lib.use_ab(False)


def _cons(memo, rows, qs, qe, k, n):
    return memo.conservation(*rows, qs, qe, k, n)


# ---------------------------------------------------------------------------------------
# golden vectors (the reference's own output), through the one-shot host ABI
# ---------------------------------------------------------------------------------------
def _dense_can_answer(rows, k, n, membership):
    """the library's own rule (memo_dense_rows_can_answer) on host columns"""
    from memo_amd.index import dense_rows_can_answer
    s, e, o = rows
    # (ALL the index's rows against its span, as query_conservation judges it: rows that can never write at k <= 64 may have
    # left the dense rows -- dense_compact -- without changing what the index can answer)
    return bool(len(s)) and int(o.min()) >= 0 and dense_rows_can_answer(len(s), int(s[0]), int(s[-1]), int(o.max()), k, n, membership)


_ONE_SHOT_SWEEPS = {}


@pytest.mark.parametrize("c", G.cases(raises=False), ids=lambda c: c["name"])
def test_golden_one_shot(c, memo, ab):
    """every golden through the one-shot seam (memo_conservation / memo_membership: host columns in, host result
    out) -- and WHICH kernel answered: wherever the dense rows alone can answer (conservation, k <= 64, <= 255 genomes,
    >= 1 row per position) the call must have gone the dense way in and run sweep_conservation_halo3_kernel, the
    benchmarked kernel (last_sweep 5): the shortest chain from the reference's own bytes to that kernel."""
    from memo_amd.index import bits_to_matrix
    rec, qs, qe = G.region(c)
    z = G.load(c)
    chrom = G.index_columns(c["index"], rec)              # whole chromosome, unfiltered
    ref = z["rows"].astype(np.int64).reshape(-1, 3)
    filtered = tuple(np.ascontiguousarray(ref[:, i]) for i in range(3))   # what filter_pq returned
    order = np.argsort(filtered[0], kind="stable")
    filtered = tuple(col[order] for col in filtered)
    for rows in (chrom, filtered):
        if c["membership"]:
            got = memo.membership(*rows, qs, qe, c["k"], c["n"])
            assert np.array_equal(bits_to_matrix(got, c["n"]), G.expected_matrix(c, z))
            text = memo.emit_membership(got, c["n"])
        else:
            got = memo.conservation(*rows, qs, qe, c["k"], c["n"])
            assert np.array_equal(got.astype(np.int64), z["vec"])
            text = memo.emit_conservation(got)
            family = ab.lib().memo_debug_last_one_shot_sweep()
            if qe > qs and c["k"] > 1 and len(rows[0]):
                dense = _dense_can_answer(rows, c["k"], c["n"], False)
                assert (family == 5) == dense, (family, dense)
                _ONE_SHOT_SWEEPS[family] = _ONE_SHOT_SWEEPS.get(family, 0) + 1
        assert G.sha(text) == c["sha256"]


@pytest.mark.parametrize("c", G.cases(raises=False)[::2], ids=lambda c: c["name"])
def test_golden_one_shot_rows(c, memo):
    """memo_conservation_rows / memo_membership_rows: the goldens' `rows` -- filter_pq's own result, uint64 [M, 3] row-major, exactly
    what memo_init is handed at memo_query.py:103 -- go in AS THEY ARE (no argsort, no column copies) and the reference's result
    comes out."""
    from memo_amd.index import bits_to_matrix
    rec, qs, qe = G.region(c)
    z = G.load(c)
    rows = z["rows"].reshape(-1, 3)
    assert rows.dtype == np.uint64
    if c["membership"]:
        got = memo.membership_rows(rows, qs, qe, c["k"], c["n"])
        assert np.array_equal(bits_to_matrix(got, c["n"]), G.expected_matrix(c, z))
        text = memo.emit_membership(got, c["n"])
    else:
        got = memo.conservation_rows(rows, qs, qe, c["k"], c["n"])
        assert np.array_equal(got.astype(np.int64), z["vec"])
        text = memo.emit_conservation(got)
    assert G.sha(text) == c["sha256"]


def test_rows_form_equals_column_form(memo, oracle):
    """the [M, 3] way in at a size where the packer's vector code, several worker tasks and both row formats are in play: equal to
    the column form and to the oracle; rows in the wrong order take the int64 way in (sorted on the device) with the same result;
    a builder fed rows in ragged pieces equals one fed columns"""
    from memo_amd import synth
    n, L = 100, 3_000_000
    num, den = synth.rows_per_position(n)
    r0, r1 = synth.shard_rows(0, L, 101, num, den, L)
    s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, n)
    rows = np.stack([s, e, o], axis=1).astype(np.uint64)
    for k in (31, 101):                                   # dense rows; 4-byte words
        want = oracle.conservation(*oracle.filter_rows(s, e, o, 5, L - 9, k), 5, L - 9, k, n, literal=False)
        assert np.array_equal(memo.conservation_rows(rows, 5, L - 9, k, n), want)
        assert np.array_equal(memo.conservation(s, e, o, 5, L - 9, k, n), want)
    wantm = oracle.membership(*oracle.filter_rows(s, e, o, 1000, 90_000, 31), 1000, 90_000, 31, n, literal=False)
    assert np.array_equal(memo.membership_rows(rows, 1000, 90_000, 31, n), wantm)
    perm = np.random.default_rng(5).permutation(len(rows))
    want = oracle.conservation(*oracle.filter_rows(s, e, o, 5, L - 9, 31), 5, L - 9, 31, n, literal=False)
    assert np.array_equal(memo.conservation_rows(rows[perm], 5, L - 9, 31, n), want)
    for dense in (True, False):
        with memo.IndexBuilder(len(rows), dense=dense) as b:
            at, step = 0, 1
            while at < len(rows):
                b.push_rows(rows[at:at + step])
                at += step
                step = step * 7 + 3
            with b.finish() as ix:
                assert np.array_equal(ix.conservation(5, L - 9, 31, n), want)


def test_golden_one_shot_reached_the_benchmarked_kernel():
    """(runs after the cases above) a fair share of the reference's goldens was answered by halo3"""
    if not _ONE_SHOT_SWEEPS:
        pytest.skip("test_golden_one_shot did not run in this session")
    assert _ONE_SHOT_SWEEPS.get(5, 0) >= 20, _ONE_SHOT_SWEEPS


@pytest.mark.parametrize("c", G.cases(raises=False)[::4], ids=lambda c: c["name"])
def test_golden_one_shot_int64_way_in(c, memo, ab):
    """the same goldens with the packed way in switched off (memo_debug_one_shot_way(1) of the AB library): int64 columns
    uploaded, finalized on the device, WideRows kernels -- what k > 256 and unpackable rows get"""
    from memo_amd.index import bits_to_matrix
    ab.check(ab.lib().memo_debug_one_shot_way(1))
    try:
        rec, qs, qe = G.region(c)
        z = G.load(c)
        rows = G.index_columns(c["index"], rec)
        if c["membership"]:
            got = memo.membership(*rows, qs, qe, c["k"], c["n"])
            assert np.array_equal(bits_to_matrix(got, c["n"]), G.expected_matrix(c, z))
        else:
            got = memo.conservation(*rows, qs, qe, c["k"], c["n"])
            assert np.array_equal(got.astype(np.int64), z["vec"])
        assert ab.lib().memo_debug_last_one_shot_sweep() in (0, 1, 7)      # (0: nothing to sweep; 1 / 7: the int64-row kernels)
    finally:
        ab.check(ab.lib().memo_debug_one_shot_way(0))


def test_goldens_on_dense_only_resident_indexes(memo):
    """every golden with n <= 255 and k <= 64 on a RESIDENT index that holds the dense rows and, as the fallback,
    the int64 columns (from_host -> pack -> pack_dense(keep_packed=False)); counts how many ran on halo3 (5) /
    planes3 (6) and how many fell back to the int64 kernels (1 / 7)."""
    from memo_amd.index import bits_to_matrix
    ran = {}
    by_index = {}
    for c in G.cases(raises=False):
        if c["n"] <= 255 and 1 < c["k"] <= 64:
            by_index.setdefault((c["index"], G.region(c)[0]), []).append(c)
    for (index, rec), cases in by_index.items():
        s, e, o = G.index_columns(index, rec)
        if not len(s) or o.max() > 255 or o.min() < 0 or s.min() < 0:
            continue
        with memo.DeviceIndex.from_host(s, e, o) as ix:
            ix.pack(keep_wide=True)
            ix.pack_dense(keep_packed=False)
            inf = ix.info()
            assert inf["dense_rows"] == 1 and inf["has_wide"] == 1
            for c in cases:
                _, qs, qe = G.region(c)
                if qe <= qs:
                    continue
                z = G.load(c)
                if c["membership"]:
                    got = ix.membership(qs, qe, c["k"], c["n"])
                    assert np.array_equal(bits_to_matrix(got, c["n"]), G.expected_matrix(c, z)), c["name"]
                else:
                    got = ix.conservation(qs, qe, c["k"], c["n"])
                    assert np.array_equal(got.astype(np.int64), z["vec"]), c["name"]
                fam = ix.info()["last_sweep"]
                ran[fam] = ran.get(fam, 0) + 1
    print("goldens on dense-only resident indexes, by kernel family:", ran)
    assert ran.get(5, 0) >= 10 and set(ran) <= {1, 5, 6, 7}, ran


@pytest.mark.parametrize("c", G.cases(raises=True), ids=lambda c: c["name"])
def test_golden_index_error(c, memo):
    rec, qs, qe = G.region(c)
    rows = G.index_columns(c["index"], rec)
    fn = memo.membership if c["membership"] else memo.conservation
    with pytest.raises(IndexError):                    # the reference's error (memo_query.py:62)
        fn(*rows, qs, qe, c["k"], c["n"])


# ---------------------------------------------------------------------------------------
# CLI-level: memo_query.main writes the reference's bytes
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [c for c in G.cases(raises=False) if c["name"].startswith("ex_")] +
                         G.cases(raises=False)[20::17], ids=lambda c: c["name"])
def test_cli_bytes(c, memo, tmp_path):
    from memo_amd import memo_query as mq
    out = tmp_path / "out.txt"
    argv = ["-b", os.path.join(G.GOLD, c["index"]), "-k", str(c["k"]), "-n", str(c["n"]),
            "-r", c["region"], "-o", str(out)] + (["-m"] if c["membership"] else [])
    mq.main(mq.parse_arguments(argv))
    data = out.read_bytes()
    assert G.sha(data) == c["sha256"]


def test_memo_front_end_subprocess(memo, tmp_path):
    """`memo query ...` as a user runs it: banner on stdout, the reference's bytes in -o."""
    import subprocess
    import sys
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "memo")
    for name in ("ex_cons_k3_0_20", "ex_memb_k3_0_20", "ex_cons_k31_full"):
        c = next(x for x in G.cases() if x["name"] == name)
        out = tmp_path / (name + ".txt")
        argv = [sys.executable, exe, "query", "-b", os.path.join(G.GOLD, c["index"]), "-n", str(c["n"]),
                "-r", c["region"], "-o", str(out)] + ([] if c["k"] == 31 else ["-k", str(c["k"])]) + \
               (["-m"] if c["membership"] else [])
        r = subprocess.run(argv, capture_output=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == (b"MEMO - membership query\n" if c["membership"] else b"MEMO - conservation query\n")
        assert out.read_bytes() == G.out_bytes(c)
    c = next(x for x in G.cases() if x["name"] == "ex_cons_n_too_small")
    r = subprocess.run([sys.executable, exe, "query", "-b", os.path.join(G.GOLD, c["index"]), "-k", "3", "-n", "2",
                        "-r", c["region"], "-o", str(tmp_path / "x.txt")], capture_output=True)
    assert r.returncode != 0 and b"IndexError" in r.stderr


def test_region_index_streams_row_groups(memo, oracle, tmp_path):
    """Parquet -> DeviceIndex row group by row group (memo_index_upload_rows + _truncate)"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    from memo_amd import memo_query as mq
    rng = np.random.default_rng(21)
    n = 30_000
    s = np.sort(rng.integers(1, 40_000, n))
    e = s + rng.integers(0, 70, n)
    o = rng.integers(1, 12, n)
    path = str(tmp_path / "rg.parquet")
    pq.write_table(pa.table({"f0": pa.array(["chrZ"] * n, pa.utf8()), "f1": s, "f2": e, "f3": o}), path,
                   row_group_size=2048, compression="ZSTD")
    for qs, qe, k in ((0, 40_000, 31), (12_345, 23_456, 31), (39_990, 40_100, 5), (50_000, 50_010, 31)):
        with mq.region_index(path, "chrZ", qs, qe + k) as ix:
            want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, 12, literal=False)
            assert np.array_equal(ix.conservation(qs, qe, k, 12), want)
            assert ix.info()["rows"] == int(((s > qs) & (s < qe + k)).sum())


# ---------------------------------------------------------------------------------------
# resident index vs oracle: many windows, every tile width, k sweep
# ---------------------------------------------------------------------------------------
def _random_index(rng, n_rows, length, n_docs, maxlen):
    s = np.sort(rng.integers(1, length, n_rows)).astype(np.int64)
    e = s + rng.integers(0, maxlen, n_rows)
    o = rng.integers(1, n_docs, n_rows).astype(np.int64)
    return s, e, o


TUNINGS = [(0, 0, 0)] + [(w, wv, 2) for w in (256, 512, 1024, 2048, 4096) for wv in (1, 4)] + \
    [(w, wv, al) for al in (3, 4) for w in (256, 512, 1024, 2048, 4096) for wv in (1, 4)]


@pytest.mark.parametrize("tile_w,waves,algo", TUNINGS)
def test_resident_index_windows(tile_w, waves, algo, memo, oracle, ab):
    """every tile width x {1, 4} waves per tile x membership {doubling, runs, planes}"""
    rng = np.random.default_rng(tile_w + 7 + waves)
    n_docs, length = 70, 60_000
    s, e, o = _random_index(rng, 250_000, length, n_docs, 140)
    if True:
        with memo.DeviceIndex.from_host(s, e, o) as ix:
            ix.debug_set_tuning(tile_w, waves, algo)
            assert ix.info()["was_sorted"] == 1
            for k in (2, 3, 4, 5, 8, 9, 16, 17, 21, 31, 32, 33, 64, 65, 101, 129, 300):
                qs = int(rng.integers(0, length // 2))
                qe = int(rng.integers(qs + 1, length + 200))
                want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                got = ix.conservation(qs, qe, k, n_docs)
                assert np.array_equal(got, want), (k, qs, qe)
                if k in (3, 31, 101):
                    qe = min(qe, qs + 9000)
                    want = oracle.membership(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                    assert np.array_equal(ix.membership(qs, qe, k, n_docs), want), (k, qs, qe)


def _fmt(n_docs):
    """the packed layout memo_index_pack / memo_builder_* choose for annots 1 .. n_docs - 1"""
    return 4 if n_docs <= 256 else (12 if n_docs <= 4096 else 6)


@pytest.mark.parametrize("n_docs,keep_wide", [(70, True), (70, False), (256, True), (257, True), (500, False), (4096, True),
                                              (4097, True), (6000, False)])
def test_packed_rows_equal_wide_rows(n_docs, keep_wide, memo, oracle, ab):
    """memo_index_pack: 4 B/row with 8-bit annots (format 4), 4 B/row with 12-bit annots and a 12-bit start
    (format 12) and 6 B/row (format 6), every tile shape, k up to 256; k > 256 falls back to the int64
    columns, or is refused when they were dropped."""
    from memo_amd import _lib
    rng = np.random.default_rng(n_docs)
    length = 150_000                                    # > 2^16: the 16-bit start wraps inside the index
    s, e, o = _random_index(rng, 400_000, length, n_docs, 300)
    e[::7] = s[::7] + rng.integers(250, 5000, len(s[::7]))        # overlaps that saturate the 8-bit length
    with memo.DeviceIndex.from_host(s, e, o) as ix:
        ix.pack(keep_wide=keep_wide)
        inf = ix.info()
        assert inf["packed_format"] == _fmt(n_docs) and inf["has_wide"] == int(keep_wide)
        try:
            for tile_w, waves, algo in [(0, 0, 0), (256, 1, 2), (512, 4, 2), (1024, 1, 2), (2048, 4, 2), (4096, 4, 0),
                                        (256, 4, 3), (2048, 4, 3), (1024, 1, 3), (0, 0, 4), (256, 1, 4), (512, 4, 4),
                                        (2048, 8, 0), (512, 8, 4),
                                        (1024, 4, 4), (2048, 4, 4), (4096, 1, 4)]:
                for k in (2, 3, 17, 31, 32, 101, 255, 256):
                    qs = int(rng.integers(0, length // 2))
                    qe = int(rng.integers(qs + 1, length + 100))
                    want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                    for scatter in (1, 2, 3):           # clipped to the tile / unclipped: doubling levels, radix-4 levels
                        ix.debug_set_tuning(tile_w, waves, algo, 0, scatter)
                        assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (k, qs, qe, tile_w, scatter)
                        if n_docs <= 255:
                            assert np.array_equal(ix.conservation(qs, qe, k, n_docs, dtype=np.uint8),
                                                  want.astype(np.uint8)), (k, qs, qe, tile_w, scatter)
                    if k in (3, 31, 256):
                        qe = min(qe, qs + 6000)
                        want = oracle.membership(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                        assert np.array_equal(ix.membership(qs, qe, k, n_docs), want), (k, qs, qe, tile_w)
        finally:
            ix.debug_set_tuning()
        if keep_wide:
            want = oracle.conservation(*oracle.filter_rows(s, e, o, 100, 9000, 300), 100, 9000, 300, n_docs, literal=False)
            assert np.array_equal(ix.conservation(100, 9000, 300, n_docs), want)
        else:
            with pytest.raises(memo.MemoError):
                ix.conservation(100, 9000, 300, n_docs)
        # num_docs too small for a row that writes: still the reference's IndexError
        with pytest.raises(IndexError):
            ix.conservation(0, length, 31, 3)


@pytest.mark.parametrize("n_docs,bucket_shift,keep_packed,keep_wide", [(100, 0, True, True), (255, 0, False, True),
                                                                      (30, 3, False, False), (300, 0, True, False),
                                                                      (90, 7, True, True)])
def test_dense_rows_equal_packed_rows(n_docs, bucket_shift, keep_packed, keep_wide, memo, oracle, ab):
    """memo_index_pack_dense: 3 bytes per row (start mod 2^10, length saturated at 63, 8-bit annot).
    The unclipped conservation sweep reads them for k <= 64, level arrays of <= 1024 cells and
    num_docs <= 511; every other query falls to the 4-byte rows / int64 columns -- or is refused when
    those were dropped.  Index longer than 2^10 and 2^16 positions: both start fields wrap inside it."""
    rng = np.random.default_rng(n_docs + bucket_shift)
    length = 200_000
    s, e, o = _random_index(rng, 700_000, length, min(n_docs, 200), 80)       # annots <= 199 also when n_docs = 300
    e[::9] = s[::9] + rng.integers(60, 300, len(s[::9]))                      # lengths that saturate 6 and 8 bits
    with memo.DeviceIndex.from_host(s, e, o, bucket_shift=bucket_shift) as ix:
        with pytest.raises(memo.MemoError):
            ix.pack_dense()                                                    # needs memo_index_pack first
        ix.pack(keep_wide=keep_wide)
        ix.pack_dense(keep_packed=keep_packed)
        inf = ix.info()
        assert inf["dense_rows"] == 1 and inf["packed_format"] == 4 and inf["has_wide"] == int(keep_wide)
        for tile_w, waves in ((0, 0), (1024, 4), (512, 1), (2048, 4), (1024, 8)):
            for k in (2, 3, 9, 17, 31, 33, 63, 64, 65, 101):
                qs = int(rng.integers(0, length // 2))
                qe = int(rng.integers(qs + 1, length + 100))
                want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                for source in (0, 3):                   # the library's choice (dense where they can answer) / the 4-byte rows
                    ix.debug_set_tuning(tile_w, waves, 0, source, 0)
                    # (256 .. 511 genomes: the table-driven kernel's nine-bit form)
                    dense_can = k <= 64 and tile_w in (0, 1024, 512) and n_docs <= 511
                    answerable = keep_packed or keep_wide or dense_can
                    if not answerable:
                        with pytest.raises(memo.MemoError):
                            ix.conservation(qs, qe, k, n_docs)
                        continue
                    assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (k, qs, qe, tile_w, source)
                    if keep_packed or dense_can:
                        # which rows answered: the dense ones wherever they can, unless the 4-byte rows were asked for
                        took_dense = ix.info()["last_sweep"] == 5
                        assert took_dense == (dense_can and not (source == 3 and keep_packed)), (k, tile_w, source)
                    if n_docs <= 255:
                        assert np.array_equal(ix.conservation(qs, qe, k, n_docs, dtype=np.uint8), want.astype(np.uint8)), \
                            (k, qs, qe, tile_w, source)
                if k in (9, 31, 33, 64, 65):
                    qe = min(qe, qs + 5000)
                    ix.debug_set_tuning(tile_w, waves)
                    dense_memb = k <= 64 and n_docs <= 255 and not keep_packed   # bit planes on the dense rows where the
                                                                                 # 4-byte rows are gone (any tile_w: it is capped)
                    if not (keep_packed or keep_wide or dense_memb):
                        with pytest.raises(memo.MemoError):
                            ix.membership(qs, qe, k, n_docs)
                    else:
                        wantb = oracle.membership(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                        assert np.array_equal(ix.membership(qs, qe, k, n_docs), wantb), (k, qs, qe, tile_w)
                        assert (ix.info()["last_sweep"] == 6) == dense_memb, (k, tile_w)
        ix.debug_set_tuning()
        ix.pack_dense(keep_packed=keep_packed)          # again: nothing to do
        if keep_wide:                                   # re-finalizing drops every packed copy
            ix.finalize(bucket_shift)
            assert ix.info()["dense_rows"] == 0 and ix.info()["packed_format"] == 0
    s2, e2, o2 = _random_index(rng, 50_000, 30_000, 700, 80)                  # annots > 511: no dense form
    with memo.DeviceIndex.from_host(s2, e2, o2) as ix:
        ix.pack()
        with pytest.raises(memo.MemoError):
            ix.pack_dense()


@pytest.mark.parametrize("bucket_shift", [1, 3, 6, 8])
def test_bucket_widths(bucket_shift, memo, oracle, ab):
    """bucket tables of 2 .. 256 positions: a tile's row slice ends at a bucket boundary, which is what
    sizes the halo of the unclipped kernels; tile widths are whole buckets"""
    from memo_amd import _lib
    rng = np.random.default_rng(40 + bucket_shift)
    n_docs, length = 90, 80_000
    s, e, o = _random_index(rng, 400_000, length, n_docs, 70)
    with memo.DeviceIndex.from_host(s, e, o, bucket_shift=bucket_shift) as ix:
        assert ix.info()["bucket_shift"] == bucket_shift
        for packed in (False, True):
            if packed:
                ix.pack(keep_wide=True)
            try:
                for tile_w, waves, algo in ((0, 0, 0), (256, 1, 3), (512, 4, 4), (2048, 4, 4), (1024, 4, 2)):
                    for k in (2, 17, 31, 32, 33, 101, 256):
                        qs = int(rng.integers(0, length // 2))
                        qe = int(rng.integers(qs + 1, length + 100))
                        rows = oracle.filter_rows(s, e, o, qs, qe, k)
                        want = oracle.conservation(*rows, qs, qe, k, n_docs, literal=False)
                        for scatter in (1, 2, 3):
                            ix.debug_set_tuning(tile_w, waves, algo, 0, scatter)
                            assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (packed, k, qs, qe, tile_w, scatter)
                        if k in (17, 31, 32, 101):
                            qe = min(qe, qs + 7000)
                            rows = oracle.filter_rows(s, e, o, qs, qe, k)
                            wantb = oracle.membership(*rows, qs, qe, k, n_docs, literal=False)
                            assert np.array_equal(ix.membership(qs, qe, k, n_docs), wantb), (packed, k, qs, qe, tile_w)
            finally:
                ix.debug_set_tuning()


# ---------------------------------------------------------------------------------------
# the packed, pinned way in (memo_builder_*): host-side narrowing + bucket table
# ---------------------------------------------------------------------------------------
def _export(ix):
    """(packed words, 16-bit annots, bucket table, rows with end < start) of a packed index"""
    import ctypes as C
    from memo_amd import _lib
    inf = ix.info()
    pk = np.empty(inf["rows"], np.uint32)
    pa = np.empty(inf["rows"] if inf["packed_format"] == 6 else 0, np.uint16)
    boff = np.empty(inf["buckets"], np.int64)
    longs = np.empty(3 * inf["long_rows"], np.int64)
    _lib.check(_lib.lib().memo_index_export_packed(ix._h, pk.ctypes.data, pa.ctypes.data if pa.size else None, boff.ctypes.data,
                                                   longs.ctypes.data if longs.size else None))
    if longs.size:                                    # (the side rows come in no particular order)
        cols = longs.reshape(3, -1)
        longs = cols[:, np.lexsort(cols[::-1])]
    return pk, pa, boff, longs


def _check_windows(ix, s, e, o, n_docs, rng, oracle, length, ks=(2, 21, 31, 64, 101, 256), windows=3):
    for k in ks:
        for _ in range(windows):
            qs = int(rng.integers(0, max(length - 10, 1)))
            qe = int(rng.integers(qs + 1, min(qs + 400_000, length + 300)))
            rows = oracle.filter_rows(s, e, o, qs, qe, k)
            want = oracle.conservation(*rows, qs, qe, k, n_docs, literal=False)
            assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (k, qs, qe)
            qe = min(qe, qs + 5000)
            rows = oracle.filter_rows(s, e, o, qs, qe, k)
            wantb = oracle.membership(*rows, qs, qe, k, n_docs, literal=False)
            assert np.array_equal(ix.membership(qs, qe, k, n_docs), wantb), (k, qs, qe)


@pytest.mark.parametrize("n_rows,n_docs,pieces", [(300_000, 90, 1), (300_000, 90, 7), (9_500_000, 200, 3),
                                                  (9_500_000, 700, 2), (40, 5, 1)])
def test_builder_equals_int64_upload(n_rows, n_docs, pieces, memo, oracle):
    """rows pushed through memo_builder_* in ragged pieces (more than one pinned slot of 4 Mi rows per
    push in the large cases) give the index memo_index_upload + _finalize + _pack give"""
    rng = np.random.default_rng(n_rows + pieces)
    length = max(n_rows // 4, 100)
    s, e, o = _random_index(rng, n_rows, length, n_docs, 90)
    e[::11] = s[::11] + rng.integers(250, 5000, len(s[::11]))      # overlaps that saturate the 8-bit length
    cuts = [0] + sorted(int(x) for x in rng.integers(0, n_rows, pieces - 1)) + [n_rows]
    with memo.IndexBuilder(n_rows + 1000) as b:
        for a, z in zip(cuts[:-1], cuts[1:]):
            b.push(s[a:z], e[a:z], o[a:z])
        b.push(s[:0], e[:0], o[:0])                                 # an empty piece changes nothing
        with b.finish() as ix:
            inf = ix.info()
            assert inf["rows"] == n_rows and inf["finalized"] == 1 and inf["has_wide"] == 0
            assert inf["packed_format"] == _fmt(n_docs)
            assert inf["min_start"] == int(s[0]) and inf["max_start"] == int(s[-1])
            with memo.DeviceIndex.from_host(s, e, o) as ref:
                ref.pack(keep_wide=True)
                assert ref.info()["buckets"] == inf["buckets"]
                # the host packer's words, 16-bit annots and bucket table are the device's, bit for bit -- once the builder's
                # rows are in the query order too (memo_index_pack on a packed index; else: once the queries that read them have made the pass worth it)
                assert inf["row_order"] == 0
                ix.pack(keep_wide=False)
                assert ix.info()["row_order"] == ref.info()["row_order"] == (0 if _fmt(n_docs) == 6 else 2)
                a, b = _export(ix), _export(ref)
                assert all(np.array_equal(x, y) for x, y in zip(a, b))
                for k in (31, 101):
                    assert np.array_equal(ix.conservation(0, length + 50, k, n_docs), ref.conservation(0, length + 50, k, n_docs))
            _check_windows(ix, s, e, o, n_docs, rng, oracle, length, windows=2)
            with pytest.raises(memo.MemoError):
                ix.conservation(0, 1000, 300, n_docs)               # k > 256 needs the int64 columns


@pytest.mark.parametrize("n_rows,n_docs,pieces", [(300_003, 90, 1), (300_001, 255, 9), (9_500_002, 200, 3), (6, 5, 1), (4, 5, 2),
                                                  (300_002, 500, 7), (300_004, 511, 2)])      # (annots of nine bits)
def test_dense_builder_equals_device_packing(n_rows, n_docs, pieces, memo, oracle, ab):
    """memo_builder_create_rows(MEMO_ROWS_DENSE): rows narrowed on the host straight to the dense format (five rows per
    16 bytes; PCIe carries 3.2 B per row), pushed in ragged pieces that end inside a group -- the groups, bucket table
    and long rows are bit for bit what memo_index_pack + memo_index_pack_dense build on the device, and the index answers
    on sweep_conservation_halo3_kernel."""
    import ctypes as C
    from memo_amd import _lib
    rng = np.random.default_rng(n_rows + pieces)
    length = max(n_rows // 4, 8)
    s, e, o = _random_index(rng, n_rows, length, n_docs, 70)
    e[::13] = s[::13] + rng.integers(60, 5000, len(s[::13]))       # overlaps that saturate the 6-bit length
    e[5::1001] = s[5::1001] - rng.integers(1, 300, len(s[5::1001]))   # a few rows with end < start
    cuts = [0] + sorted(int(x) for x in rng.integers(0, n_rows, pieces - 1)) + [n_rows]

    def export(ix):
        inf = ix.info()
        g = np.empty(4 * ((inf["dense_row_count"] + 4) // 5), np.uint32)      # (rows that cannot write at k <= 64 may be left out)
        boff = np.empty(inf["buckets"], np.int64)
        longs = np.empty(3 * inf["long_rows"], np.int64)
        _lib.check(_lib.lib().memo_index_export_dense(ix._h, g.ctypes.data, boff.ctypes.data, longs.ctypes.data if len(longs) else None))
        return g, boff, longs
    with memo.IndexBuilder(n_rows + 77, dense=True) as b:
        for a, z in zip(cuts[:-1], cuts[1:]):
            b.push(s[a:z], e[a:z], o[a:z])
        with b.finish() as ix:
            inf = ix.info()
            assert inf["rows"] == n_rows and inf["dense_rows"] == 1 and inf["has_wide"] == 0 and inf["finalized"] == 1
            assert inf["device_bytes"] < 3.3 * n_rows + 16 * inf["buckets"] + 2_000_000
            with memo.DeviceIndex.from_host(s, e, o) as ref:
                ref.debug_row_order(1)        # (the device's groups in START order, as the host builder emits them: the 4-byte
                ref.pack(keep_wide=False)     #  words they are made from are otherwise dealt over their buckets' starts)
                ref.pack_dense(keep_packed=False)
                got, want = export(ix), export(ref)
                assert inf["dense_row_count"] == ref.info()["dense_row_count"]
                pad = (5 - inf["dense_row_count"] % 5) % 5         # rows behind the last one in its group: never read by number
                if pad == 0:
                    assert np.array_equal(got[0], want[0])
                else:
                    assert np.array_equal(got[0][:-4], want[0][:-4])
                assert np.array_equal(got[1], want[1])

                def triples(x):                                   # (the device collects them with atomics: any order)
                    t = x.reshape(3, -1).T
                    return t[np.lexsort(t.T[::-1])]
                assert np.array_equal(triples(got[2]), triples(want[2]))
            if n_rows > 1000:
                for k in (31, 64, 2):
                    qs, qe = int(rng.integers(0, length // 2)), int(rng.integers(length // 2, length + 40))
                    want_v = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                    assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want_v), (k, qs, qe)
                    assert ix.info()["last_sweep"] == 5
                qe2 = min(qe, qs + 20_000)
                if n_docs <= 255:
                    wantb = oracle.membership(*oracle.filter_rows(s, e, o, qs, qe2, 31), qs, qe2, 31, n_docs, literal=False)
                    assert np.array_equal(ix.membership(qs, qe2, 31, n_docs), wantb) and ix.info()["last_sweep"] == 6
                else:                                                # (bit planes on the dense rows: up to 255 genomes)
                    with pytest.raises(memo.MemoError):
                        ix.membership(qs, qe2, 31, n_docs)
                with pytest.raises(memo.MemoError):
                    ix.conservation(0, 1000, 101, n_docs)            # k > 64 needs the 4-byte rows
    # an annot above 511 is refused: the caller starts over with the 4-byte rows
    o2 = o.copy()
    o2[-1] = 512
    with memo.IndexBuilder(n_rows, dense=True) as b:
        with pytest.raises(memo.MemoUnpackable):
            b.push(s, e, o2)
        with pytest.raises(memo.MemoError):
            b.finish()
    # import of a slice that starts inside a group (what the cache does), straight through the ABI
    if n_rows > 100_000:
        with memo.DeviceIndex.from_host(s, e, o) as ref:
            ref.pack(keep_wide=False)
            ref.pack_dense(keep_packed=True)
            g, boff, longs = export(ref)
            inf = ref.info()
            shift, nb = inf["bucket_shift"], inf["buckets"]
            for qs, qe, k in ((length // 3 + 3, length // 2, 31), (1, 777, 64), (length - 500, length + 10, 5)):
                b_lo = min(max(qs, 0) >> shift, nb - 1)
                b_hi = min(((qe + k) >> shift) + 1, nb - 1)
                r0, r1 = int(boff[b_lo]), int(boff[b_hi])
                base = r0 // 5 * 5
                table = np.ascontiguousarray(boff[b_lo:b_hi + 1])
                grp = np.ascontiguousarray(g[4 * (base // 5):4 * ((r1 + 4) // 5)])
                h = C.c_void_p()
                _lib.check(_lib.lib().memo_index_import_dense(r1 - base, 0, shift, b_lo, grp.ctypes.data, table.ctypes.data,
                                                              len(table) + 1, base, max(b_lo << shift, int(s[0])),
                                                              min(((b_hi + 1) << shift) - 1, int(s[-1])), inf["max_annot"],
                                                              longs.ctypes.data if len(longs) else None, len(longs) // 3, C.byref(h)))
                with memo.DeviceIndex(r1 - base, 0, _handle=h) as part:
                    want_v = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                    assert np.array_equal(part.conservation(qs, qe, k, n_docs), want_v), (qs, qe, k)
                    assert part.info()["last_sweep"] == 5 and part.info()["bucket_base"] == b_lo


@pytest.mark.parametrize("way", ["device", "builder"])
def test_dense_rows_leave_out_rows_that_cannot_write(way, memo, oracle, ab):
    """An index built from sequences has 40 % of its rows with an overlap of 63 and more (profiles/r03_realistic_index*):
    such a row -- like a row with end < start -- can never write at k <= 64, which is all the dense rows answer, so the
    dense rows leave them out when they are more than a tenth (dense_compact): their own row numbers, their own bucket
    table, fewer bytes to sweep.  Same results from every kernel that reads them, slices included."""
    import ctypes as C
    from memo_amd import _lib
    rng = np.random.default_rng(41)
    n_rows, length, n_docs = 2_000_003, 400_000, 90
    s, e, o = _random_index(rng, n_rows, length, n_docs, 62)
    far = rng.random(n_rows) < 0.45
    e[far] = s[far] + rng.integers(63, 4000, int(far.sum()))          # 45 % of the rows: overlaps of 63 .. 4000
    e[7::5003] = s[7::5003] - rng.integers(1, 200, len(s[7::5003]))    # and a few rows with end < start
    clump = (s > 100_000) & (s < 100_300)                              # a stretch where nearly every row goes
    e[clump] = s[clump] + 500
    if way == "device":
        ix = memo.DeviceIndex.from_host(s, e, o)
        ix.pack(keep_wide=False)
        ix.pack_dense(keep_packed=False)
    else:
        cuts = [0, 3, 700_001, 1_234_567, n_rows]
        with memo.IndexBuilder(n_rows, dense=True) as b:
            for a, z in zip(cuts[:-1], cuts[1:]):
                b.push(s[a:z], e[a:z], o[a:z])
            ix = b.finish()
    with ix:
        inf = ix.info()
        kept = int(((e - s >= 0) & (e - s < 63)).sum())
        assert inf["rows"] == n_rows and inf["dense_row_count"] == kept and inf["dense_rows"] == 1
        assert inf["device_bytes"] < 3.3 * kept + 8 * inf["buckets"] * 2 + 3_000_000
        for k in (31, 64, 5, 21):
            for qs, qe in ((0, length + 30), (99_000, 102_004), (4, 333_336), (123_457, 300_001)):
                want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                for src in (0, 5):                                        # the table-driven kernel, the round-2 kernel
                    ix.debug_set_tuning(0, 0, 0, src, 0)
                    assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (k, qs, qe, src)
                    assert ix.info()["last_sweep"] == 5
            ix.debug_set_tuning(0, 0, 0, 0, 0)
            wantb = oracle.membership(*oracle.filter_rows(s, e, o, 90_000, 120_000, k), 90_000, 120_000, k, n_docs, literal=False)
            assert np.array_equal(ix.membership(90_000, 120_000, k, n_docs), wantb) and ix.info()["last_sweep"] == 6
        # export / import of a slice of the (shorter) dense stream with its own table
        g = np.empty(4 * ((kept + 4) // 5), np.uint32)
        boff3 = np.empty(inf["buckets"], np.int64)
        longs = np.empty(3 * inf["long_rows"], np.int64)
        _lib.check(_lib.lib().memo_index_export_dense(ix._h, g.ctypes.data, boff3.ctypes.data, longs.ctypes.data))
        assert boff3[-1] == kept and np.all(np.diff(boff3) >= 0)
        shift, nb = inf["bucket_shift"], inf["buckets"]
        for qs, qe, k in ((150_003, 250_000, 31), (99_000, 101_000, 64)):
            b_lo, b_hi = qs >> shift, min(((qe + k) >> shift) + 1, nb - 1)
            d0, d1 = int(boff3[b_lo]), int(boff3[b_hi])
            base = d0 // 5 * 5
            table = np.ascontiguousarray(boff3[b_lo:b_hi + 1])
            grp = np.ascontiguousarray(g[4 * (base // 5):4 * ((d1 + 4) // 5)])
            h = C.c_void_p()
            _lib.check(_lib.lib().memo_index_import_dense(d1 - base, 0, shift, b_lo, grp.ctypes.data, table.ctypes.data, len(table) + 1,
                                                          base, b_lo << shift, ((b_hi + 1) << shift) - 1, inf["max_annot"],
                                                          longs.ctypes.data, len(longs) // 3, C.byref(h)))
            with memo.DeviceIndex(d1 - base, 0, _handle=h) as part:
                want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                assert np.array_equal(part.conservation(qs, qe, k, n_docs), want), (qs, qe, k)


def test_builder_switches_to_12_bit_annots_late(memo, oracle):
    """the first annot > 255 arrives after 5 M rows are already on the device in format 4: they are rewritten
    in format 12 in place (widen_annot_kernel) and the current piece is packed again"""
    rng = np.random.default_rng(5)
    n_rows, length, n_docs = 6_000_000, 900_000, 1000
    s, e, o = _random_index(rng, n_rows, length, n_docs, 70)
    o[:5_000_000] = rng.integers(1, 200, 5_000_000)
    with memo.IndexBuilder(n_rows) as b:
        for a in range(0, n_rows, 1_000_000):
            b.push(s[a:a + 1_000_000], e[a:a + 1_000_000], o[a:a + 1_000_000])
        with b.finish() as ix:
            assert ix.info()["packed_format"] == 12
            _check_windows(ix, s, e, o, n_docs, rng, oracle, length, ks=(31, 101), windows=3)
            ix.pack()                                               # (a packed index: its rows into the query order now)
            with memo.DeviceIndex.from_host(s, e, o) as ref:        # the device's format-12 words are the same words
                ref.pack()
                assert all(np.array_equal(x, y) for x, y in zip(_export(ix), _export(ref)))


def test_builder_refuses_what_cannot_be_packed(memo, oracle):
    rng = np.random.default_rng(6)
    s, e, o = _random_index(rng, 50_000, 20_000, 30, 80)
    bad = {"unsorted": (s[::-1].copy(), e[::-1].copy(), o), "negative start": (s - 30_000, e - 30_000, o),
           "annot 70000": (s, e, np.where(np.arange(len(o)) == 777, 70_000, o)),
           "annot 5000": (s, e, np.where(np.arange(len(o)) == 777, 5000, o)),
           "negative annot": (s, e, np.where(np.arange(len(o)) == 40_000, -1, o))}
    for name, (bs, be, bo) in bad.items():
        with memo.IndexBuilder(len(bs)) as b:
            with pytest.raises(memo.MemoUnpackable):
                b.push(bs, be, bo)
            with pytest.raises(memo.MemoError):
                b.push(s, e, o)                                     # a failed builder stays failed
        if name == "annot 70000":
            continue                                                # (IndexError in the reference: covered elsewhere)
        # the one-shot seam falls back to the int64 way in and still equals the reference's result
        order = np.argsort(bs, kind="stable")
        n_docs = 30 if name not in ("negative annot", "annot 5000") else (31 if name == "negative annot" else 5001)
        qs, qe = (-25_000, -5_000) if name == "negative start" else (1000, 15_000)
        want = oracle.conservation(*oracle.filter_rows(bs[order], be[order], bo[order], qs, qe, 31), qs, qe, 31, n_docs,
                                   literal=False)
        assert np.array_equal(memo.conservation(bs, be, bo, qs, qe, 31, n_docs), want), name
    # sorted across pieces, too
    with memo.IndexBuilder(len(s)) as b:
        b.push(s[25_000:], e[25_000:], o[25_000:])
        with pytest.raises(memo.MemoUnpackable):
            b.push(s[:25_000], e[:25_000], o[:25_000])
    with memo.IndexBuilder(10) as b:
        with pytest.raises(memo.MemoError):
            b.push(s[:11], e[:11], o[:11])                          # more rows than announced


def test_builder_rows_with_end_before_start_and_empty(memo, oracle):
    rng = np.random.default_rng(8)
    s, e, o = _random_index(rng, 200_000, 60_000, 40, 70)
    neg = rng.random(len(s)) < 0.01
    e[neg] = s[neg] - rng.integers(1, 3000, int(neg.sum()))
    with memo.DeviceIndex.from_host_packed(s, e, o) as ix:
        with memo.DeviceIndex.from_host(s, e, o) as ref:
            ref.pack()
            assert ix.info()["long_rows"] == ref.info()["long_rows"] == int(neg.sum())
            ix.pack(keep_wide=False)                                # (the builder's rows into the query order: memo_interleave.hip)
            assert all(np.array_equal(x, y) for x, y in zip(_export(ix), _export(ref)))
        _check_windows(ix, s, e, o, 40, rng, oracle, 60_000, ks=(3, 31, 200), windows=3)
        for L in (1, 2, 3, 5, 7):                                   # result tails shorter than a 32-bit word
            want = oracle.conservation(*oracle.filter_rows(s, e, o, 100, 100 + L, 31), 100, 100 + L, 31, 40, literal=False)
            assert np.array_equal(ix.conservation(100, 100 + L, 31, 40), want)
            assert np.array_equal(ix.conservation(100, 100 + L, 31, 40, dtype=np.uint8), want.astype(np.uint8))
    with memo.IndexBuilder(0) as b:
        with b.finish() as ix:
            assert ix.info()["rows"] == 0
            assert np.array_equal(ix.conservation(5, 50, 31, 9), np.full(45, 9, np.uint16))
            assert np.array_equal(ix.membership(5, 8, 31, 9), np.full((3, 1), 511, np.uint32))


# ---------------------------------------------------------------------------------------
# the CLI's sidecar cache of packed rows (memo_amd/cache.py, memo_index_export_packed / _import_packed)
# ---------------------------------------------------------------------------------------
def test_sidecar_cache_round_trip(memo, oracle, tmp_path, monkeypatch):
    import subprocess
    import sys
    import time
    import pyarrow as pa
    import pyarrow.parquet as pq
    from memo_amd import cache, memo_query as mq
    rng = np.random.default_rng(12)
    tabs, cols = [], {}
    for name, n, n_docs in (("chrA", 700_000, 60), ("chr B/2", 250_000, 700), ("chrC", 450_000, 400)):   # formats 4, 12 and 12 with annots of nine bits
        s, e, o = _random_index(rng, n, 150_000, n_docs, 80)
        neg = rng.random(n) < 0.001
        e[neg] = s[neg] - rng.integers(1, 500, int(neg.sum()))                        # a few rows with end < start
        cols[name] = (s, e, o, n_docs)
        tabs.append(pa.table({"f0": pa.array([name] * n, pa.utf8()), "f1": s, "f2": e, "f3": o}))
    path = str(tmp_path / "idx.parquet")
    pq.write_table(pa.concat_tables(tabs), path, row_group_size=50_000, compression="ZSTD")
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "memo")

    def query(record, qs, qe, k, n_docs, mode, extra=()):
        out = tmp_path / "out.txt"
        r = subprocess.run([sys.executable, exe, "query", "-b", path, "-k", str(k), "-n", str(n_docs), "-r",
                            f"{record}:{qs}-{qe}", "-o", str(out), *extra], capture_output=True,
                           env=dict(os.environ, MEMO_CACHE=mode, MEMO_TIMING="1"))
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        return out.read_bytes(), r.stderr.decode()

    for record, (s, e, o, n_docs) in cols.items():
        assert not os.path.exists(cache.cache_path(path, record))
        text0, _ = query(record, 1000, 60_000, 31, n_docs, "0")                     # no cache read or written
        assert not os.path.exists(cache.cache_path(path, record))
        text1, err1 = query(record, 1000, 60_000, 31, n_docs, "sync")               # miss: answered from Parquet, cache written
        assert text1 == text0 and "sidecar cache" not in err1
        assert os.path.exists(cache.cache_path(path, record))
        text2, err2 = query(record, 1000, 60_000, 31, n_docs, "1")                  # hit
        assert text2 == text0 and "from the sidecar cache, ctypes-only path" in err2
        # the cache holds the dense rows where every annot fits 9 bits; a conservation query they can answer reads THEM
        assert ("dense rows (3.2 B)" if n_docs <= 511 else "4-byte rows") in err2, err2
        assert "dense rows (3.2 B)" in err1 if n_docs <= 511 else "4-byte rows" in err1, err1     # ... and so does the miss
        if n_docs <= 511:
            # v3: the miss ran with -k 31, so the file also holds that class's VIEW of the dense rows (overlaps below 30: three
            # eighths of these rows) and a hit whose k fits the class uploads and sweeps only those
            head = cache._open(path, record)[0]
            assert head["view"]["cap"] == 30 and 0.3 < head["view"]["rows"] / head["rows3"] < 0.55     # (rows3: what is left after dense_compact)
            assert "the k-class view" in err2, err2
            import re

            def rows_of(err):
                return int(re.search(r"\((\d+) rows as", err).group(1))
            rows_hit = rows_of(err2)
            text21, err21 = query(record, 1000, 60_000, 21, n_docs, "1")              # k - 1 = 20 <= 30: the same view serves it
            assert "the k-class view" in err21
            text33, err33 = query(record, 1000, 60_000, 33, n_docs, "1")              # k - 1 = 32 > 30: all the dense rows
            assert "dense rows (3.2 B)" in err33 and "the k-class view" not in err33
            assert rows_of(err33) > 1.8 * rows_hit
            for kk, tx in ((21, text21), (33, text33)):
                assert tx == memo.emit_conservation(oracle.conservation(*oracle.filter_rows(s, e, o, 1000, 60_000, kk), 1000, 60_000, kk,
                                                                        n_docs, literal=False))
        text2w, err2w = query(record, 1000, 60_000, 300, n_docs, "1")               # k > 256: not the fast path, not the cache
        assert "sidecar cache" not in err2w
        assert text2w == memo.emit_conservation(oracle.conservation(*oracle.filter_rows(s, e, o, 1000, 60_000, 300), 1000,
                                                                    60_000, 300, n_docs, literal=False))
        for region, nd in ((f"{record}:500-100", n_docs), (f"{record}:0-150100", 3), (f"{record}:200000-200010", n_docs),
                           (f"{record}:7-7", n_docs), ("nochr:5-50", n_docs)):      # errors and empty answers: the same
            got = []                                                                 # from either path
            for mode in ("0", "read"):
                out = tmp_path / "out_e.txt"
                if out.exists():
                    out.unlink()
                r = subprocess.run([sys.executable, exe, "query", "-b", path, "-k", "31", "-n", str(nd), "-r", region, "-o",
                                    str(out)], capture_output=True, env=dict(os.environ, MEMO_CACHE=mode))
                last = r.stderr.decode().strip().splitlines()[-1:] if r.returncode else []
                got.append((r.returncode, last, out.read_bytes() if out.exists() else None))
            assert got[0] == got[1], (region, got)
        # the fast path imports neither NumPy nor Arrow
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        code = ("import sys; sys.path.insert(0, %r); from memo_amd import _fastquery as f; "
                "assert f.try_query(%r, %r, '31', %r, %r, False); "
                "assert 'numpy' not in sys.modules and 'pyarrow' not in sys.modules"
                % (root, path, f"{record}:1000-60000", str(n_docs), str(tmp_path / "out_f.txt")))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=dict(os.environ, MEMO_CACHE="read"))
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        assert (tmp_path / "out_f.txt").read_bytes() == text0
        textm, errm = query(record, 70_000, 71_000, 21, n_docs, "read", ("-m",))    # membership from the cache, too
        assert "from the sidecar cache" in errm
        assert textm == memo.emit_membership(oracle.membership(*oracle.filter_rows(s, e, o, 70_000, 71_000, 21), 70_000,
                                                               71_000, 21, n_docs, literal=False), n_docs)
        monkeypatch.setenv("MEMO_CACHE", "read")
        for qs, qe, k in ((0, 150_100, 31), (149_000, 160_000, 101), (77_777, 77_778, 2), (5, 6, 256), (200_000, 200_010, 31),
                          (31, 64, 31), (32, 95, 3), (1003, 90_001, 64), (64, 65_000, 21)):
            with mq.region_index(path, record, qs, qe + k, k=k) as ix:
                assert ix.cache == "hit"
                want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (record, qs, qe, k)
            # with the query known, the slice comes in as dense rows wherever they can answer (row slices that start
            # inside a group of five, bucket tables rebased to the group's first row)
            with mq.region_index(path, record, qs, qe + k, k=k, num_docs=n_docs, membership=False) as ix:
                assert ix.cache == "hit"
                assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (record, qs, qe, k)
                inf = ix.info()
                if n_docs <= 511 and k <= 64 and inf["rows"] >= inf["max_start"] - inf["min_start"] + 1 and inf["rows"]:
                    assert inf["dense_rows"] == 1 and inf["last_sweep"] == 5, (inf, k)
                else:
                    assert inf["dense_rows"] == 0
        with mq.region_index(path, record, 10, 5000 + 300, k=300) as ix:            # k > 256: int64 columns, not the cache
            assert ix.cache is None and ix.info()["has_wide"] == 1
    # a changed index file invalidates its caches
    time.sleep(0.01)
    pq.write_table(tabs[0].slice(0, 200_000), path, row_group_size=50_000, compression="ZSTD")
    s, e, o, n_docs = cols["chrA"]
    with mq.region_index(path, "chrA", 0, 50_000 + 31, k=31) as ix:
        assert ix.cache == (path, "chrA")                                          # stale: ignored
        want = oracle.conservation(*oracle.filter_rows(s[:200_000], e[:200_000], o[:200_000], 0, 50_000, 31), 0, 50_000, 31,
                                   n_docs, literal=False)
        assert np.array_equal(ix.conservation(0, 50_000, 31, n_docs), want)
    assert cache.build(path, "chrA") and cache.build(path, "nochr") is None
    with mq.region_index(path, "chrA", 0, 50_000 + 31, k=31) as ix:
        assert ix.cache == "hit" and np.array_equal(ix.conservation(0, 50_000, 31, n_docs), want)


# ---------------------------------------------------------------------------------------
# several GPUs from one process (include/memo_amd_multi.h).  One GPU here: the device list names it
# several times, which runs every code path (threads, partition, peer copies into the root's result)
# ---------------------------------------------------------------------------------------
def test_multi_device_host_form(memo, oracle):
    from memo_amd import index
    rng = np.random.default_rng(31)
    n_docs, length = 60, 300_000
    s, e, o = _random_index(rng, 1_200_000, length, n_docs, 90)
    for devices in ([0], [0, 0], [0, 0, 0, 0, 0]):
        for k, qs, qe in ((31, 0, length + 50), (101, 12_345, 250_001), (2, 299_000, 299_013), (300, 5, 200_000)):
            want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
            assert np.array_equal(index.conservation_multi(s, e, o, qs, qe, k, n_docs, devices), want), (devices, k, qs)
            qe2 = min(qe, qs + 40_000)
            wantb = oracle.membership(*oracle.filter_rows(s, e, o, qs, qe2, k), qs, qe2, k, n_docs, literal=False)
            assert np.array_equal(index.membership_multi(s, e, o, qs, qe2, k, n_docs, devices), wantb), (devices, k, qs)
    # unsorted rows fall back to one device (which sorts); errors of a sub-window reach the caller
    perm = rng.permutation(len(s))
    want = oracle.conservation(*oracle.filter_rows(s, e, o, 0, 100_000, 31), 0, 100_000, 31, n_docs, literal=False)
    assert np.array_equal(index.conservation_multi(s[perm], e[perm], o[perm], 0, 100_000, 31, n_docs, [0, 0]), want)
    with pytest.raises(IndexError):
        index.conservation_multi(s, e, o, 0, length, 31, 5, [0, 0, 0])
    with pytest.raises(ValueError):
        index.conservation_multi(s, e, o, 10, 5, 31, n_docs, [0, 0])
    with pytest.raises(memo.MemoError):
        index.conservation_multi(s, e, o, 0, 1000, 31, n_docs, [0, 7])          # no such device


@pytest.mark.parametrize("root_weight", [1.0, 0.5, 0.0])
def test_multi_device_resident_form(root_weight, memo, oracle):
    import ctypes as C
    from memo_amd import index, _lib
    rng = np.random.default_rng(32)
    n_docs, length = 60, 400_000
    s, e, o = _random_index(rng, 1_500_000, length, n_docs, 90)
    shards = [memo.DeviceIndex.from_host_packed(s, e, o) for _ in range(3)]      # replicas ("one per GPU")
    try:
        for membership in (False, True):
            for k, qs, qe in ((31, 0, length), (101, 777, 333_333), (64, 100_000, 100_100)):
                if membership:
                    qe = min(qe, qs + 50_000)
                    want = oracle.membership(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                    got = np.empty_like(want)
                else:
                    want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                    got = np.empty_like(want)
                d = C.c_void_p()
                _lib.check(_lib.lib().memo_dev_malloc(0, max(got.nbytes, 16), C.byref(d)))
                try:
                    index.query_multi_dev(shards, qs, qe, k, n_docs, d.value, 0, None, root_weight, membership)
                    for ix in shards:
                        ix.check()
                    _lib.check(_lib.lib().memo_dev_download(0, got.ctypes.data, d, got.nbytes, None))
                finally:
                    _lib.lib().memo_dev_free(0, d)
                assert np.array_equal(got, want), (membership, k, qs, qe)
    finally:
        for ix in shards:
            ix.close()


def test_multi_device_rows_with_end_before_start(memo, oracle):
    """A row with end < start shades [end - (k-1), start): any distance LEFT of its start, across sub-window cuts.
    The rule 'sub-window [a, b) needs the rows a < start < b + k' does not hold for it (round-2 ADVICE): the host
    form must not split such an index, the resident form filters those rows by the whole window."""
    import ctypes as C
    from memo_amd import index, _lib
    rng = np.random.default_rng(33)
    n_docs, length = 40, 200_000
    s, e, o = _random_index(rng, 600_000, length, n_docs, 90)
    # 300 rows reaching 100 .. 150 000 positions left of their start -- across one or several cuts
    pick = rng.choice(len(s), 300, replace=False)
    e[pick] = np.maximum(s[pick] - rng.integers(100, 150_000, 300), -5)
    mid = length // 2
    j = int(np.searchsorted(s, mid + 41))          # the ADVICE example: start just right of a cut + k, end far left of it
    e[j] = s[j] - 400
    for k, qs, qe in ((31, 0, length), (101, 1_000, 180_001), (5, 50_000, 150_000)):
        want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
        qe2 = min(qe, qs + 60_000)
        wantb = oracle.membership(*oracle.filter_rows(s, e, o, qs, qe2, k), qs, qe2, k, n_docs, literal=False)
        for devices in ([0, 0], [0, 0, 0, 0, 0, 0, 0]):
            assert np.array_equal(index.conservation_multi(s, e, o, qs, qe, k, n_docs, devices), want), (devices, k)
            assert np.array_equal(index.membership_multi(s, e, o, qs, qe2, k, n_docs, devices), wantb), (devices, k)
        shards = [memo.DeviceIndex.from_host_packed(s, e, o) for _ in range(4)]
        try:
            assert shards[0].info()["long_rows"] >= 300
            for membership, w, a, b in ((False, want, qs, qe), (True, wantb, qs, qe2)):
                for root_weight in (1.0, 0.3):
                    got = np.empty_like(w)
                    d = C.c_void_p()
                    _lib.check(_lib.lib().memo_dev_malloc(0, max(got.nbytes, 16), C.byref(d)))
                    try:
                        index.query_multi_dev(shards, a, b, k, n_docs, d.value, 0, None, root_weight, membership)
                        for ix in shards:
                            ix.check()
                        _lib.check(_lib.lib().memo_dev_download(0, got.ctypes.data, d, got.nbytes, None))
                    finally:
                        _lib.lib().memo_dev_free(0, d)
                    assert np.array_equal(got, w), (membership, k, root_weight)
            # the single-window calls on the same index are untouched by the sub-window state
            assert np.array_equal(shards[0].conservation(qs, qe, k, n_docs), want)
        finally:
            for ix in shards:
                ix.close()


def test_integration_stub_from_the_docs(memo, oracle):
    """the ctypes stub a maintainer of the reference would add, executed as it is
    printed there (only the library path filled in), against reference goldens"""
    import re
    from memo_amd import _lib
    doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    block = re.search(r"## 2\. The stub.*?```python\n(.*?)```", doc, re.S).group(1)
    ns = {}
    exec(block.replace("/path/to/libmemo_amd.so", _lib.SO_PATH), ns)
    for c in [c for c in G.cases(raises=False) if c["name"].startswith("ex_")][:6] + G.cases(raises=False)[50:60]:
        rec, qs, qe = G.region(c)
        z = G.load(c)
        arr = z["rows"].reshape(-1, 3).astype(np.uint64)                      # what filter_pq returned
        res = ns["memo_query_gpu"](arr, c["k"], qs, qe, c["n"], c["membership"])
        if c["membership"]:
            g = np.arange(c["n"])
            got = ((res[:, g >> 5] >> (g & 31).astype(np.uint32)) & 1).astype("byte")
            assert np.array_equal(got, G.expected_matrix(c, z))
        else:
            assert np.array_equal(res.astype(np.int64), z["vec"])


def test_two_threads_two_indexes(memo, oracle):
    """one index per thread.  No mutable state is shared between indexes (the
    kernel-shape choices are per index, the error message is thread-local), so two threads querying two
    different indexes at once -- one of them raising the reference's IndexError every other query --
    get their own results and their own errors."""
    import threading
    rng = np.random.default_rng(77)
    data = [_random_index(rng, 120_000, 50_000, nd, 90) for nd in (40, 300)]
    n_docs = (40, 300)
    errors = []

    def worker(t):
        try:
            s, e, o = data[t]
            r = np.random.default_rng(t)
            with memo.DeviceIndex.from_host(s, e, o) as ix:
                if t == 1:
                    ix.pack(keep_wide=True)
                for i in range(40):
                    k = int(r.choice([3, 21, 31, 101]))
                    qs = int(r.integers(0, 30_000))
                    qe = qs + int(r.integers(1, 20_000))
                    if t == 0 and i % 2:
                        with pytest.raises(IndexError):
                            ix.conservation(0, 50_000, 31, 5)        # annots up to 39 do not fit 6 columns
                    rows = oracle.filter_rows(s, e, o, qs, qe, k)
                    want = oracle.conservation(*rows, qs, qe, k, n_docs[t], literal=False)
                    assert np.array_equal(ix.conservation(qs, qe, k, n_docs[t]), want), (t, i, k, qs, qe)
                    qe = min(qe, qs + 4000)
                    rows = oracle.filter_rows(s, e, o, qs, qe, k)
                    wantb = oracle.membership(*rows, qs, qe, k, n_docs[t], literal=False)
                    assert np.array_equal(ix.membership(qs, qe, k, n_docs[t]), wantb), (t, i, k, qs, qe)
        except BaseException as exc:       # noqa: BLE001 -- reported by the main thread
            errors.append((t, repr(exc)))

    th = [threading.Thread(target=worker, args=(t,)) for t in (0, 1)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors


def test_ragged_density_and_edges(memo, oracle):
    """clumped starts, empty stretches, window beyond the last row, window before the first."""
    rng = np.random.default_rng(99)
    parts = [rng.integers(1, 50, 4000), rng.integers(5000, 5003, 3000), rng.integers(20000, 90000, 500),
             np.full(2000, 123456)]
    s = np.sort(np.concatenate(parts)).astype(np.int64)
    e = s + rng.integers(0, 70, len(s))
    o = rng.integers(1, 9, len(s)).astype(np.int64)
    with memo.DeviceIndex.from_host(s, e, o) as ix:
        for qs, qe in ((0, 130000), (0, 1), (49, 5003), (4990, 5010), (123400, 123500), (123456, 123460),
                       (200000, 200100), (5, 5), (122880, 124928), (1023, 1025)):
            for k in (1, 2, 31, 64):
                want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, 9, literal=False)
                assert np.array_equal(ix.conservation(qs, qe, k, 9), want), (qs, qe, k)
                want = oracle.membership(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, 9, literal=False)
                assert np.array_equal(ix.membership(qs, qe, k, 9), want), (qs, qe, k)


def test_reversed_window_raises_like_the_reference(memo, oracle):
    s = np.array([5], np.int64)
    for fn in (memo.conservation, memo.membership, oracle.conservation, oracle.membership, oracle.np_conservation):
        with pytest.raises(ValueError):                # np.zeros([negative, ...]) in memo_init
            fn(s, s + 1, s * 0 + 1, 7, 3, 31, 5)


def test_empty_index_and_empty_window(memo):
    z = np.zeros(0, np.int64)
    assert np.array_equal(memo.conservation(z, z, z, 10, 20, 31, 5), np.full(10, 5, np.uint16))
    assert memo.conservation(z, z, z, 10, 10, 31, 5).size == 0
    m = memo.membership(z, z, z, 0, 3, 31, 40)
    assert m.shape == (3, 2) and np.all(m[:, 0] == 0xFFFFFFFF) and np.all(m[:, 1] == 0xFF)


def test_num_docs_word_boundaries(memo, oracle):
    rng = np.random.default_rng(5)
    for n_docs in (2, 31, 32, 33, 64, 65, 200, 500):
        s, e, o = _random_index(rng, 20000, 3000, n_docs, 60)
        want = oracle.membership(s, e, o, 100, 2900, 31, n_docs, literal=False)
        assert np.array_equal(memo.membership(s, e, o, 100, 2900, 31, n_docs), want), n_docs
        want = oracle.conservation(s, e, o, 100, 2900, 31, n_docs, literal=False)
        assert np.array_equal(memo.conservation(s, e, o, 100, 2900, 31, n_docs), want), n_docs


@pytest.mark.parametrize("n_docs", [20, 50, 100, 250, 257, 500])
def test_membership_planes_by_result_words(n_docs, memo, oracle):
    """the planes kernels at the tile width the launcher derives from the number of result words (4096 / words positions, whole
    32-position words, whole buckets: 1, 2, 4, 8, 9 and 16 words here), both instantiations -- runs of two words at most (k - 1 <= 31: the
    hand-written row block, a second word only where the run reaches it) and longer ones (whole words as plain stores) -- on the
    4-byte rows and on the dense rows, bucket widths 1, 2 and 32; windows that start and end inside a tile
    (reference: src/memo_query.py:50-51, :60-62)"""
    rng = np.random.default_rng(500 + n_docs)
    length = 40_000
    s, e, o = _random_index(rng, 150_000, length, n_docs, 120)
    for bucket_shift in (0, 1, 5):
        for dense in (False, True):
            if dense and n_docs > 255:
                continue
            with memo.DeviceIndex.from_host(s, e, o, bucket_shift=bucket_shift) as ix:
                ix.pack()
                if dense:
                    ix.pack_dense(keep_packed=False)
                    assert ix.info()["dense_rows"] == 1
                for k in (2, 8, 31, 32, 33, 64, 101, 200):
                    if dense and k > 64:
                        continue
                    qs = int(rng.integers(0, length // 3))
                    qe = int(rng.integers(qs + 1, min(qs + 9000, length + 50)))
                    rows = oracle.filter_rows(s, e, o, qs, qe, k)
                    want = oracle.membership(*rows, qs, qe, k, n_docs, literal=False)
                    assert np.array_equal(ix.membership(qs, qe, k, n_docs), want), (bucket_shift, dense, k, qs, qe)
                    assert ix.info()["last_sweep"] == (6 if dense else 7), (dense, k, ix.info()["last_sweep"])
                ix.check()


def test_membership_many_genomes_is_sliced(memo, oracle):
    """num_docs in the thousands: the genome words do not fit one LDS tile and are swept in slices"""
    rng = np.random.default_rng(12)
    for n_docs in (2049, 5000, 20000):
        s, e, o = _random_index(rng, 30000, 2000, n_docs, 60)
        want = oracle.membership(s, e, o, 100, 1500, 31, n_docs, literal=False)
        with memo.DeviceIndex.from_host(s, e, o) as ix:
            assert np.array_equal(ix.membership(100, 1500, 31, n_docs), want), n_docs
            ix.pack()
            assert np.array_equal(ix.membership(100, 1500, 31, n_docs), want), n_docs
            assert np.array_equal(ix.conservation(100, 1500, 31, n_docs),
                                  oracle.conservation(s, e, o, 100, 1500, 31, n_docs, literal=False))


def test_annot_edge_values(memo, oracle):
    """order 0, order == N (the sentinel column), negative order (NumPy wraps), order > N (IndexError)."""
    s = np.array([10, 20, 30, 40], np.int64)
    e = s + 2
    for o, ok in (([0, 5, 4, 3], True), ([5, 5, 5, 5], True), ([-1, -6, 2, 1], True), ([1, 6, 1, 1], False),
                  ([1, -7, 1, 1], False)):
        o = np.array(o, np.int64)
        if ok:
            assert np.array_equal(memo.conservation(s, e, o, 0, 50, 8, 5), oracle.conservation(s, e, o, 0, 50, 8, 5))
        else:
            with pytest.raises(IndexError):
                memo.conservation(s, e, o, 0, 50, 8, 5)
            with pytest.raises(IndexError):
                oracle.conservation(s, e, o, 0, 50, 8, 5)
    # the offending row is outside the window / does not write: no error, as in the reference
    o = np.array([1, 99, 1, 1], np.int64)
    assert np.array_equal(memo.conservation(s, e, o, 25, 50, 8, 5), oracle.conservation(s, e, o, 25, 50, 8, 5))


def test_unsorted_rows_are_sorted_on_device(memo, oracle):
    from memo_amd import _lib
    rng = np.random.default_rng(11)
    s, e, o = _random_index(rng, 50000, 20000, 12, 80)
    perm = rng.permutation(len(s))
    want = oracle.conservation(s, e, o, 0, 20000, 31, 12, literal=False)
    with memo.DeviceIndex.from_host(s[perm], e[perm], o[perm]) as ix:
        assert ix.info()["was_sorted"] == 0
        assert np.array_equal(ix.conservation(0, 20000, 31, 12), want)
    with pytest.raises(memo.MemoError) as ei:
        memo.DeviceIndex.from_host(s[perm], e[perm], o[perm], allow_sort=False)
    assert ei.value.code == _lib.MEMO_EUNSORTED


def test_rows_with_end_before_start(memo, oracle):
    """end < start never comes out of dap_to_bed.py, but memo_query.py accepts it and such a row can
    shade far more than k-1 positions; a side pass applies those rows after the sweep."""
    rng = np.random.default_rng(17)
    n_docs, length = 40, 30_000
    s, e, o = _random_index(rng, 60_000, length, n_docs, 90)
    neg = rng.random(len(s)) < 0.03
    e[neg] = s[neg] - rng.integers(1, 5000, int(neg.sum()))
    assert np.array_equal(memo.conservation([5, 9], [7, 8], [1, 1], 0, 20, 3, 5),
                          oracle.conservation([5, 9], [7, 8], [1, 1], 0, 20, 3, 5))
    with memo.DeviceIndex.from_host(s, e, o) as ix:
        for packed in (False, True):
            if packed:
                ix.pack(keep_wide=True)
            for k in (1, 2, 31, 101, 300):
                for qs, qe in ((0, length), (7000, 9000), (123, 124), (length - 50, length + 400)):
                    rows = oracle.filter_rows(s, e, o, qs, qe, k)
                    want = oracle.conservation(*rows, qs, qe, k, n_docs, literal=False)
                    assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (packed, k, qs, qe)
                    assert np.array_equal(ix.conservation(qs, qe, k, n_docs, dtype=np.uint8), want.astype(np.uint8))
                    if qe - qs <= 5000:
                        wantb = oracle.membership(*rows, qs, qe, k, n_docs, literal=False)
                        assert np.array_equal(ix.membership(qs, qe, k, n_docs), wantb), (packed, k, qs, qe)
        with pytest.raises(IndexError):                       # a long row with a column outside the matrix
            ix.conservation(0, length, 31, 5)


# ---------------------------------------------------------------------------------------
# BASELINE configs on the synthetic pangenome
# ---------------------------------------------------------------------------------------
def test_config2_full_window(memo, oracle):
    """config 2: 10 genomes x 10 Mbp, ~5 M rows, k=31 -- whole window against the oracle."""
    from memo_amd import synth
    n, L, k = 10, 10_000_000, 31
    ix, (r0, r1) = synth.device_index(0, L, k, n, L)
    num, den = synth.rows_per_position(n)
    s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, n)
    with ix:
        got = ix.conservation(0, L, k, n)
        assert np.array_equal(got, oracle.conservation(s, e, o, 0, L, k, n, literal=False))
        sub = ix.conservation(1_234_567, 2_345_678, k, n)
        assert np.array_equal(sub, got[1_234_567:2_345_678])
        for kk in (21, 101):
            assert np.array_equal(ix.conservation(0, 2_000_000, kk, n),
                                  oracle.conservation(s, e, o, 0, 2_000_000, kk, n, literal=False))
        m = ix.membership(0, 3_000_000, k, n)
        assert np.array_equal(m, oracle.membership(s, e, o, 0, 3_000_000, k, n, literal=False))


# config 3 conservation whole-window FNV of the closed-form oracle (checksum of per-chunk checksums,
# oracle.synth_window_compare): every instantiation below must reproduce it
_C3_FNV = {}


@pytest.mark.parametrize("membership,pack,dtype,k", [
    (False, None, np.uint16, 31), (False, "keep", np.uint16, 31), (False, "keep", np.uint8, 31), (False, "only", np.uint8, 31),
    (False, "dense", np.uint8, 31), (False, "dense", np.uint16, 31), (True, None, None, 31), (True, "keep", None, 31),
    (False, "only", np.uint8, 101), (False, "keep", np.uint16, 101), (False, "only", np.uint8, 128), (True, "only", None, 101),
    (True, "dense", None, 31)],
    ids=["cons-int64-u16", "cons-packed-u16", "cons-packed-u8", "cons-packedonly-u8", "cons-dense-u8", "cons-dense-u16",
         "memb-int64", "memb-packed", "cons-packedonly-u8-k101-mixed", "cons-packed-u16-k101-mixed", "cons-packedonly-u8-k128-mixed2",
         "memb-packedonly-k101", "memb-dense-planes3"])
def test_config3_full_size_properties(membership, pack, dtype, k, memo, oracle):
    """configs 3/4 at full size: 100 genomes x 100 Mbp, 500 M rows, on the int64 columns AND on the
    packed rows (the benchmarked kernels: sweep_conservation_halo_kernel<PackedRows<false,false>,..,uint8>,
    sweep_membership_planes_kernel), uint16 and uint8 results.  The WHOLE result is compared with the
    closed-form oracle (chunks regenerate their rows; all host cores), its checksum of checksums must be
    the same for every instantiation, and the split-window property (any sub-window query == slice of the
    full query) is checked on sampled sub-windows."""
    import ctypes as C
    from memo_amd import synth, _lib
    n, L = 100, 100_000_000
    ix, (r0, r1) = synth.device_index(0, L, k, n, L, pack=pack)
    W = (n + 31) // 32
    with ix:
        inf = ix.info()
        assert inf["rows"] == r1 - r0 and abs((r1 - r0) - 500_000_000) < 1000 + 5 * k
        assert inf["packed_format"] == (4 if pack else 0) and inf["has_wide"] == (0 if pack in ("only", "dense") else 1)
        assert inf["dense_rows"] == int(pack == "dense")
        full = np.empty((L, W), np.uint32) if membership else np.empty(L, dtype)

        def whole_window(into, times=1):
            d = C.c_void_p()
            _lib.check(_lib.lib().memo_dev_malloc(0, into.nbytes, C.byref(d)))
            try:
                for _ in range(times):
                    if membership:
                        ix.membership_dev(0, L, k, n, d.value)
                    elif dtype == np.uint8:
                        ix.conservation_u8_dev(0, L, k, n, d.value)
                    else:
                        ix.conservation_dev(0, L, k, n, d.value)
                ix.check()
                _lib.check(_lib.lib().memo_dev_download(0, into.ctypes.data, d, into.nbytes, None))
            finally:
                _lib.lib().memo_dev_free(0, d)

        whole_window(full)
        assert ix.info()["last_rows_read"] == r1 - r0          # (the first query of a class reads all the rows)
        bad, fnv = oracle.synth_window_compare(full, 0, L, k, n, L, membership=membership)
        assert bad == 0, f"{bad} chunks of the whole-window result differ from the oracle"
        assert _C3_FNV.setdefault((membership, k), fnv) == fnv
        if not membership and pack in ("keep", "only"):   # the kernel family the library chose
            assert ix.info()["last_sweep"] == {31: 2, 101: 4, 128: 4}[k]
            if k > 64:        # the level plan: only the arrays some row can write to (k = 101: blocks of 64 and 32 + the fold's
                assert ix.info()["last_level_arrays"] == {101: 3, 128: 2}[k]    # blocks of 16; k = 128: blocks of 64 + those)
        if pack == "dense":                               # a fact, not an inference: halo3 / planes3 answered
            assert ix.info()["last_sweep"] == (6 if membership else 5)
        rng = np.random.default_rng(3)
        for a in [0, L - 300_000] + [int(x) for x in rng.integers(0, L - 300_000, 4)]:
            b = a + 300_000
            sub = ix.membership(a + 17, b - 5, k, n) if membership else ix.conservation(a + 17, b - 5, k, n, dtype=dtype)
            assert np.array_equal(sub, full[a + 17:b - 5]), a
        if not membership:
            assert full.min() >= 1 and full.max() == n
        if pack and k == 31:
            # THE SCOREBOARD INSTANTIATION: what bench.py times is the sweep on the k-class VIEW of the
            # rows -- memo_index_prepare builds it, as bench.py does (queries build it once it has become worth its pass:
            # test_views_are_built_when_they_have_paid_for_themselves) -- read through the tile table.  The whole window again,
            # and THAT result against the oracle over all 10^8 positions (not a sub-window, not an inference from the all-rows
            # result).  Dense rows, conservation: the library's choice is the view of SIX rows per group (round 5; asserted), and the
            # view of five (MEMO_OPT_VIEW_ROWS 5: bench.py's other_row_formats line) is checked the same way.
            for view_rows in ((0, 5) if pack == "dense" and not membership else (0,)):
                ix.set_option(4, view_rows)
                ix.prepare(k, n, membership)
                again = np.empty_like(full)
                whole_window(again, times=1)
                inf = ix.info()
                six = pack == "dense" and not membership and view_rows == 0
                assert inf["last_rows_read"] < r1 - r0 and abs(inf["last_rows_read"] / (r1 - r0) - (0.516 if six else 0.5)) < 0.01, inf
                assert inf["views_resident"] >= 1
                if pack == "dense" and not membership:
                    assert inf["last_sweep"] == 5 and inf["last_variant"] == (3 if six else 2) and inf["tile_tables_resident"] >= 1, inf
                bad2, fnv2 = oracle.synth_window_compare(again, 0, L, k, n, L, membership=membership)
                assert bad2 == 0 and fnv2 == fnv, f"{bad2} chunks of the view's whole-window result differ from the oracle ({view_rows})"


# ---------------------------------------------------------------------------------------
# `memo view` binning on the device (plot_conservation.py:46-65)
# ---------------------------------------------------------------------------------------
def test_view_binning_matches_reference(memo, oracle):
    import json
    from memo_amd import view
    for c in json.load(open(os.path.join(G.GOLD, "view", "manifest.json"))):
        z = np.load(os.path.join(G.GOLD, "view", c["name"] + ".npz"))
        if "raises" in c:
            with pytest.raises(ZeroDivisionError):
                view.preprocess_data(z["vec"], c["n_docs"], c["n_bins"])
            continue
        got = view.preprocess_data(z["vec"], c["n_docs"], c["n_bins"])
        assert np.array_equal(got["bin"], z["bin"]) and np.array_equal(got["No. Genomes"], z["genomes"])
        assert np.array_equal(got["value"], z["value"])      # float64, bit for bit
    rng = np.random.default_rng(2)
    for n_docs, n_bins, npos in ((100, 500, 3_000_000), (20000, 3, 100_000), (7, 1, 1)):
        vec = rng.integers(0, n_docs + 1, npos).astype(np.uint16)
        got = view.preprocess_data(vec, n_docs, n_bins)
        want = oracle.view_table(vec, n_docs, n_bins) if n_docs < 1000 else None
        if want:
            assert np.array_equal(got["value"], want["value"])
        counts, edges = view.bin_counts(vec, n_docs, n_bins)
        assert counts.sum() == npos and np.array_equal(counts.sum(0), np.bincount(vec, minlength=n_docs + 1))


def test_view_from_text_file(memo, tmp_path):
    from memo_amd import view
    vec = np.array([5, 5, 3, 4, 5, 2, 1, 2, 5, 5, 4, 4], np.uint16)
    p = tmp_path / "c.txt"
    p.write_bytes(memo.emit_conservation(vec))
    t = view.preprocess_data(str(p), 5, 3)
    assert np.allclose(t["value"].reshape(5, 3)[:, 0], [0, 0, 0, 0.25, 0.25])


def test_cli_sharded_path_single_rank(memo, tmp_path):
    """the torch.distributed form of `memo query` (RCCL initialised, shard + gather code path) with
    one rank; N > 1 needs more GPUs than this box has and is covered on CPU by test_shard_gloo.py"""
    import subprocess
    import sys
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bin", "memo")
    for name in ("ex_cons_k3_0_20", "ex_memb_k3_0_20", "rnd_n40_cons_k31_c0w0"):
        c = next(x for x in G.cases() if x["name"] == name)
        out = tmp_path / (name + ".txt")
        argv = [sys.executable, exe, "query", "-b", os.path.join(G.GOLD, c["index"]), "-n", str(c["n"]), "-k", str(c["k"]),
                "-r", c["region"], "-o", str(out)] + (["-m"] if c["membership"] else [])
        import socket
        for attempt in range(2):
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            r = subprocess.run(argv, capture_output=True,
                               env=dict(os.environ, MEMO_FORCE_SHARDED="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
            err = r.stderr.decode(errors="replace")
            # (one run of the round-6 tier failed here and passed on the next two boxes: a rendezvous that does not come up --
            # the port taken between our bind and the child's, RCCL's bootstrap -- is tried once more; anything else fails at once)
            if r.returncode == 0 or not any(w in err for w in ("NCCL", "RCCL", "Address already in use", "Connection", "rendezvous", "store")):
                break
        assert r.returncode == 0, (name, err[-3000:])
        assert G.sha(out.read_bytes()) == c["sha256"]


def test_more_than_2_pow_32_rows(memo, oracle, ab):
    """4.4e9 rows (106 GB of int64 columns + 18 GB packed in HBM): row numbers, byte offsets and
    pivot coordinates beyond 32 bits; windows at both ends against the oracle."""
    from memo_amd import synth
    n, k = 20, 31                                   # 1 row per pivot position
    pivot = 4_400_000_000
    num, den = synth.rows_per_position(n)
    ix, (r0, r1) = synth.device_index(0, pivot, k, n, pivot, pack="keep")
    with ix:
        assert r1 - r0 > 2 ** 32 and ix.info()["rows"] == r1 - r0 and ix.info()["packed_format"] == 4
        for source in (0, 1):                       # packed rows, then the int64 columns
            ix.debug_set_tuning(row_source=source)
            if True:
                for a in (0, 2 ** 32 - 70_000, pivot - 150_000):
                    b = min(a + 140_001, pivot + 100)
                    sr0, sr1 = synth.shard_rows(a, b, k, num, den, pivot)
                    s, e, o = oracle.synth_rows(sr0, sr1 - sr0, num, den, n)
                    want = oracle.conservation(s, e, o, a, b, k, n, literal=False)
                    assert np.array_equal(ix.conservation(a, b, k, n), want), (source, a)
                    wantb = oracle.membership(s, e, o, a, a + 3000, k, n, literal=False)
                    assert np.array_equal(ix.membership(a, a + 3000, k, n), wantb), (source, a)


# ---------------------------------------------------------------------------------------
# index-row construction on the device (dap_to_bed.py:55-134)
# ---------------------------------------------------------------------------------------
def _dap_cases():
    import json
    return json.load(open(os.path.join(G.GOLD, "dap", "manifest.json")))


@pytest.mark.parametrize("c", _dap_cases(), ids=lambda c: c["name"])
def test_dap_to_bed_matches_reference(c, memo):
    import io
    from memo_amd import dap_to_bed as D
    d = os.path.join(G.GOLD, "dap")
    argv = ["--mem", "--fai", os.path.join(d, c["fai"]), "--dap", os.path.join(d, c["dap"])] + \
        (["--overlap"] if c["overlap"] else []) + (["--order"] if c["order"] else [])
    args = D.parse_arguments(argv)
    D.check_args(args)
    buf = io.BytesIO()
    D.main(args, buf)
    assert G.sha(buf.getvalue()) == c["sha256"]
    assert buf.getvalue() == open(os.path.join(d, c["name"] + ".bed"), "rb").read()


def test_dap_streaming_chunks_and_truncated_record(memo):
    """rows pushed in many small pieces == one push; a DAP that stops inside a record still gets its
    chr-end rows (dap_to_bed.py:133-134); checked against the oracle restatement"""
    from memo_amd.dap_to_bed import DapConverter
    from oracle import dap_oracle as O
    rng = np.random.default_rng(4)
    rec_begin = np.array([0, 700, 701, 1500, 4000], np.int64)
    for C_, npos, order, overlap in ((7, 4000, True, True), (130, 3333, False, True), (600, 900, True, True), (33, 4000, False, False)):
        lcp = rng.integers(0, 40, (npos, C_)).astype(np.int32)
        want = O.dap_rows(lcp, rec_begin, overlap, order)
        if npos < rec_begin[-1]:                   # truncated: the oracle sees the stream end as a record end
            pass
        for pieces in (1, 7, 64):
            with DapConverter(C_, rec_begin, order, overlap) as conv:
                got = [conv.push(part) for part in np.array_split(lcp, pieces)] + [conv.finish()]
            cat = [np.concatenate([g[i] for g in got]) for i in range(4)]
            assert all(np.array_equal(a, b) for a, b in zip(cat, want)), (C_, npos, pieces)


def test_dap_to_parquet_is_a_queryable_index(memo, oracle, tmp_path):
    """DAP -> Parquet on the device, then `memo query` on it == oracle on the oracle's rows"""
    from memo_amd import dap_to_bed as D, memo_query as mq
    from oracle import dap_oracle as O
    d = os.path.join(G.GOLD, "dap")
    out = str(tmp_path / "idx.parquet")
    D.dap_to_parquet(os.path.join(d, "dap_wide.dap.txt"), os.path.join(d, "dap_wide.fa.fai"), out, order=True)
    names, rec_begin = O.read_fai(os.path.join(d, "dap_wide.fa.fai"))
    _, lcp = O.read_dap(os.path.join(d, "dap_wide.dap.txt"))
    rec, s, e, a = O.dap_rows(lcp, rec_begin, True, True)
    sel = rec == 0
    want = oracle.conservation(*oracle.filter_rows(s[sel], e[sel], a[sel], 10, 290, 5), 10, 290, 5, 71, literal=False)
    with mq.region_index(out, names[0], 10, 290 + 5) as ix:
        assert np.array_equal(ix.conservation(10, 290, 5, 71), want)


# ---------------------------------------------------------------------------------------
# randomized differential test and API error paths
# ---------------------------------------------------------------------------------------
def test_randomized_differential(memo, oracle):
    """many small random indexes / windows / k / N through the resident index (both row formats)"""
    rng = np.random.default_rng(20261003)
    for case in range(120):
        n_docs = int(rng.choice([2, 3, 9, 32, 33, 100, 255, 256, 300]))
        length = int(rng.choice([40, 300, 5000, 70000]))
        m = int(rng.integers(0, 4000))
        style = case % 4
        if style == 0:
            s = np.sort(rng.integers(1, length, m))
        elif style == 1:                                  # heavy clumps: many rows on few starts
            s = np.sort(rng.choice(rng.integers(1, length, 5), m))
        elif style == 2:                                  # starts beyond / before the window
            s = np.sort(rng.integers(-50, length + 500, m))
        else:                                             # runs of consecutive starts
            s = np.sort(np.repeat(rng.integers(1, length, m // 8 + 1), 8)[:m] + rng.integers(0, 3, m))
        s = s.astype(np.int64)
        e = s + rng.integers(0, int(rng.choice([3, 40, 400])), m)
        o = rng.integers(1, n_docs, m).astype(np.int64) if n_docs > 1 else np.ones(m, np.int64)
        k = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 101, 127, 128, 129, 255, 256, 257, 600]))
        qs = int(rng.integers(0, length))
        qe = int(rng.integers(qs, length + 100))
        rows = oracle.filter_rows(s, e, o, qs, qe, k)
        want = oracle.conservation(*rows, qs, qe, k, n_docs, literal=False)
        wantb = oracle.membership(*rows, qs, qe, k, n_docs, literal=False)
        with memo.DeviceIndex.from_host(s, e, o) as ix:
            for packed in (False, True):
                if packed:
                    if s.size and s.min() < 0:
                        with pytest.raises(memo.MemoError):
                            ix.pack()
                        break
                    ix.pack(keep_wide=True)
                assert np.array_equal(ix.conservation(qs, qe, k, n_docs), want), (case, packed)
                assert np.array_equal(ix.membership(qs, qe, k, n_docs), wantb), (case, packed)
                if n_docs <= 255:
                    assert np.array_equal(ix.conservation(qs, qe, k, n_docs, dtype=np.uint8), want.astype(np.uint8)), case


def test_api_error_paths(memo):
    import ctypes as C
    from memo_amd import _lib
    L = _lib.lib()
    s = np.arange(1, 101, dtype=np.int64)
    ix = memo.DeviceIndex(100, 0)
    try:
        _lib.check(L.memo_index_upload(ix._h, s.ctypes.data, (s + 2).ctypes.data, (s * 0 + 1).ctypes.data, 100))
        d = C.c_void_p()
        _lib.check(L.memo_dev_malloc(0, 4096, C.byref(d)))
        assert L.memo_query_conservation_dev(ix._h, 0, 50, 31, 5, d, None) == _lib.MEMO_ENOTREADY
        ix.finalize()
        assert L.memo_query_conservation_dev(ix._h, 0, 50, 31, 0, d, None) == _lib.MEMO_EINVAL          # num_docs
        assert L.memo_query_conservation_dev(ix._h, 0, 50, 31, 70000, d, None) == _lib.MEMO_EINVAL
        assert L.memo_query_conservation_u8_dev(ix._h, 0, 50, 31, 300, d, None) == _lib.MEMO_EINVAL      # uint8 needs N <= 255
        assert L.memo_query_conservation_dev(ix._h, 0, 50, 31, 5, C.c_void_p(d.value + 2), None) == _lib.MEMO_EINVAL  # alignment
        assert L.memo_query_conservation_dev(ix._h, 0, 50, 31, 5, None, None) == _lib.MEMO_EINVAL
        assert L.memo_query_conservation_dev(ix._h, 50, 0, 31, 5, d, None) == _lib.MEMO_EINVAL
        assert b"ValueError" in L.memo_last_error()
        assert L.memo_query_conservation_dev(ix._h, 0, 0, 31, 5, None, None) == 0                         # empty window
        assert L.memo_index_upload(ix._h, s.ctypes.data, s.ctypes.data, s.ctypes.data, 99) == _lib.MEMO_EINVAL
        assert L.memo_index_pack(ix._h, 0) == 0
        assert L.memo_index_upload(ix._h, s.ctypes.data, s.ctypes.data, s.ctypes.data, 100) == _lib.MEMO_EINVAL  # columns dropped
        assert L.memo_query_conservation_dev(ix._h, 0, 50, 300, 5, d, None) == _lib.MEMO_EINVAL           # k > 256 needs int64 rows
        assert L.memo_query_conservation_dev(ix._h, 0, 50, 31, 5, d, None) == 0
        ix.check()
        _lib.check(L.memo_dev_free(0, d))
        bad = C.c_void_p()
        assert L.memo_index_create(10, 99, C.byref(bad)) == _lib.MEMO_EHIP
    finally:
        ix.close()


def test_config2_size_numpy_rng_index(memo, oracle):
    """a NumPy-RNG index at config-2 size (random starts: ragged density, ties),
    whole window, both row formats, against the oracle's closed form"""
    rng = np.random.default_rng(0x4D454D4F)
    n, L, m, k = 10, 10_000_000, 5_000_000, 31
    s = np.sort(rng.integers(1, L, m)).astype(np.int64)
    e = s + rng.integers(0, 60, m)
    o = rng.integers(1, n, m).astype(np.int64)
    want = oracle.conservation(s, e, o, 0, L, k, n, literal=False)
    with memo.DeviceIndex.from_host(s, e, o) as ix:
        assert np.array_equal(ix.conservation(0, L, k, n), want)
        ix.pack(keep_wide=False)
        assert np.array_equal(ix.conservation(0, L, k, n), want)
        wantb = oracle.membership(*oracle.filter_rows(s, e, o, 4_000_000, 5_000_000, k), 4_000_000, 5_000_000, k, n, literal=False)
        assert np.array_equal(ix.membership(4_000_000, 5_000_000, k, n), wantb)


def test_config5_shard_packed_rows(memo, oracle):
    """one shard of config 5 (500 genomes, 25 rows per position, 2^25 positions, 8.4e8 rows): packed
    4-byte rows with 12-bit annots only (the int64 columns dropped, as an HPRC-scale deployment would), k in {21, 31, 101},
    sampled sub-windows against the oracle + the split-window property"""
    from memo_amd import synth
    n, L = 500, 1 << 25
    pivot = 8 * L                                    # this is shard 3 of 8
    qs, qe = 3 * L, 4 * L
    num, den = synth.rows_per_position(n)
    ix, (r0, r1) = synth.device_index(qs, qe, 101, n, pivot, pack="only")
    with ix:
        inf = ix.info()
        assert inf["packed_format"] == 12 and inf["has_wide"] == 0 and inf["rows"] == r1 - r0
        assert inf["device_bytes"] < 5 * (r1 - r0)
        rng = np.random.default_rng(9)
        for k in (21, 31, 101):
            full = ix.conservation(qs, qe, k, n)
            assert full.dtype == np.uint16 and 1 <= full.min() and full.max() <= n
            assert ix.info()["last_rows_read"] == r1 - r0 or k != 21
            bad, fnv = oracle.synth_window_compare(full, qs, qe, k, n, pivot)   # the WHOLE 2^25-position shard
            assert bad == 0, (k, bad)
            if k in (21, 31):     # the benchmarked regime: the k-class view of the words (memo_index_prepare builds it), whole shard
                ix.prepare(k, n)
                again = ix.conservation(qs, qe, k, n)
                inf = ix.info()
                assert inf["last_rows_read"] < r1 - r0 and abs(inf["last_rows_read"] / (r1 - r0) - (k - 1) / 60) < 0.01, (k, inf)
                bad2, fnv2 = oracle.synth_window_compare(again, qs, qe, k, n, pivot)
                assert bad2 == 0 and fnv2 == fnv, (k, bad2)
            assert ix.info()["row_order"] == 2      # (memo_index_pack dealt the rows over their buckets' starts)
            for a in [qs, qe - 200_000] + [int(x) for x in rng.integers(qs, qe - 200_000, 3)]:
                b = a + 200_000
                sr0, sr1 = synth.shard_rows(a, b, k, num, den, pivot)
                s, e, o = oracle.synth_rows(sr0, sr1 - sr0, num, den, n)
                want = oracle.conservation(s, e, o, a, b, k, n, literal=False)
                assert np.array_equal(full[a - qs:b - qs], want), (k, a)
                assert np.array_equal(ix.conservation(a + 3, b - 11, k, n), want[3:-11])


def test_config5_shard_dense_rows(memo, oracle):
    """the same shard of config 5 on the DENSE rows (500 genomes: nine-bit annots, memo_sweep_cons3t.hip A9): what `bench.py
    --workload c5 --k 21 / 31` times since round 4.  The whole 2^25-position shard against the oracle before and after the
    k-class view exists, a window that begins off a multiple of four, sampled sub-windows, k = 64 (every row, six level arrays)."""
    from memo_amd import synth
    n, L = 500, 1 << 25
    pivot = 8 * L
    qs, qe = 3 * L, 4 * L
    num, den = synth.rows_per_position(n)
    ix, (r0, r1) = synth.device_index(qs, qe, 64, n, pivot, pack="dense")
    with ix:
        inf = ix.info()
        assert inf["dense_rows"] == 1 and inf["has_wide"] == 0 and inf["max_annot"] > 255
        assert inf["device_bytes"] < 3.3 * (r1 - r0) + (64 << 20)
        rng = np.random.default_rng(10)
        for k in (21, 31, 64):
            full = ix.conservation(qs, qe, k, n)
            inf = ix.info()
            assert (inf["last_sweep"], inf["last_variant"]) == (5, 2) and full.dtype == np.uint16 and full.max() <= n
            bad, fnv = oracle.synth_window_compare(full, qs, qe, k, n, pivot)   # the WHOLE shard, all the dense rows
            assert bad == 0, (k, bad)
            if k != 64:           # the benchmarked regime: the k-class view (memo_index_prepare builds it), whole shard
                ix.prepare(k, n)
                again = ix.conservation(qs, qe, k, n)
                inf = ix.info()
                assert inf["last_rows_read"] < 0.6 * (r1 - r0) and (inf["last_sweep"], inf["last_variant"]) == (5, 2), (k, inf)
                bad2, fnv2 = oracle.synth_window_compare(again, qs, qe, k, n, pivot)
                assert bad2 == 0 and fnv2 == fnv, (k, bad2)
                off = ix.conservation(qs + 1, qe - 2, k, n)                     # off the 4-position raster: the same kernel
                assert ix.info()["last_variant"] == 2 and np.array_equal(off, full[1:-2])
            for a in [qs, qe - 200_000] + [int(x) for x in rng.integers(qs, qe - 200_000, 2)]:
                b = a + 200_000
                sr0, sr1 = synth.shard_rows(a, b, k, num, den, pivot)
                s, e, o = oracle.synth_rows(sr0, sr1 - sr0, num, den, n)
                want = oracle.conservation(s, e, o, a, b, k, n, literal=False)
                assert np.array_equal(full[a - qs:b - qs], want), (k, a)
                assert np.array_equal(ix.conservation(a + 3, b - 11, k, n), want[3:-11])
        ix.check()


def test_dense_row_sweep_variants(memo, oracle, ab):
    """The dense rows are swept by sweep_conservation_halo3t_kernel (the tile's row slice from the index's tile table;
    memo_sweep_cons3t.hip) wherever the query fits it; the round-2 kernel (every wave works its tile out) answers the
    rest.  Both, every
    k <= 64 that changes the number of level arrays, windows on and off the 4-position raster, both result types:
    bit-equal to each other and to the oracle; the table is rebuilt when the dense rows change; five values of k make the
    four-table cache evict."""
    from memo_amd import synth
    n, L = 100, 6_000_000
    ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack="dense")
    num, den = synth.rows_per_position(n)
    with ix:
        ix.set_option(3, 100000)  # MEMO_OPT_BUILD_COST_PCT: no view in this part (sixty queries of k = 31's class are about what one costs;
                                  # a view of six rows would answer as variant 3)
        for k in (2, 3, 4, 5, 8, 9, 16, 17, 21, 31, 32, 33, 48, 64, 31):
            for qs, qe in ((0, L), (4, L - 3), (1_000_000, 5_000_001), (2_345_676, 2_345_680 + 1_500_000), (777, 5_555_555)):
                for dt in (np.uint8, np.uint16):
                    ix.debug_set_tuning(0, 0, 0, 5, 0)
                    ref = ix.conservation(qs, qe, k, n, dtype=dt)
                    assert ix.info()["last_sweep"] == 5 and ix.info()["last_variant"] == 0
                    for src, variant in ((8, 2), (0, 2)):
                        ix.debug_set_tuning(0, 0, 0, src, 0)
                        got = ix.conservation(qs, qe, k, n, dtype=dt)
                        assert np.array_equal(got, ref), (src, k, qs, qe, dt)
                        inf = ix.info()
                        assert inf["last_sweep"] == 5
                        assert inf["last_variant"] == variant, (src, k, qs, qe, inf)  # (the table kernel takes every window, off the 4-position raster too)
            a, b = 3_000_000, 3_300_000
            sr0, sr1 = synth.shard_rows(a, b, k, num, den, L)
            s, e, o = oracle.synth_rows(sr0, sr1 - sr0, num, den, n)
            ix.debug_set_tuning(0, 0, 0, 0, 0)
            assert np.array_equal(ix.conservation(a, b, k, n), oracle.conservation(s, e, o, a, b, k, n, literal=False)), k
    # k-class views: conservation with k - 1 <= 2 / 4 / 6 ... / 32 reads the dense rows whose overlap is below that cap when that
    # spares a fifth of them (here: overlaps uniform in 0 .. 59, so every class does); row_source 9 reads them all
    ix, (r0, r1) = synth.device_index(0, 2_000_000, 64, n, 2_000_000, pack="dense")
    with ix:
        ix.set_option(3, 0)       # MEMO_OPT_BUILD_COST_PCT 0: the first query of a class builds its view
        ix.set_option(4, 5)       # MEMO_OPT_VIEW_ROWS 5: five rows per group (the views of six: test_six_row_views)
        for k, cap in ((2, 2), (5, 4), (6, 6), (9, 8), (10, 10), (17, 16), (18, 18), (21, 20), (22, 22), (30, 30), (31, 30), (32, 32),
                       (33, 32), (34, None), (64, None)):
            ix.debug_set_tuning(0, 0, 0, 9, 0)
            ref = ix.conservation(0, 2_000_000, k, n, dtype=np.uint8)
            inf = ix.info()
            assert inf["last_rows_read"] == inf["dense_row_count"] == r1 - r0
            for src in (0, 10):
                ix.debug_set_tuning(0, 0, 0, src, 0)
                for _ in range(2 if src == 0 else 1):
                    assert np.array_equal(ix.conservation(0, 2_000_000, k, n, dtype=np.uint8), ref), (k, src)
                inf = ix.info()
                if cap is None:
                    assert inf["last_rows_read"] == r1 - r0
                else:
                    assert abs(inf["last_rows_read"] / (r1 - r0) - cap / 60) < 0.01, (k, inf["last_rows_read"])
    # a window that runs far past the last row, and one that starts past it: tiles beyond the table read its last (empty) entry
    rng = np.random.default_rng(77)
    s, e, o = _random_index(rng, 700_000, 200_000, 60, 70)
    with memo.DeviceIndex.from_host_packed(s, e, o, dense=True) as ix2:
        for qs, qe in ((0, 1_000_000), (150_000, 3_000_000), (400_000, 900_000), (199_996, 200_100)):
            for k in (31, 5):
                want = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, 60, literal=False)
                assert np.array_equal(ix2.conservation(qs, qe, k, 60), want), (qs, qe, k)
                assert ix2.info()["last_variant"] == 2


@pytest.mark.parametrize("n", [100, 500])
def test_packed_k_class_views(n, memo, oracle, ab):
    """The 4-byte words have k-class views too (packed_rows_for: the rows whose overlap is below the class's cap -- 2 ... 32 by 2, ... 64 by 8, ... 128 by 16 --, built by
    the query that finds it worth its pass, when that spares a fifth of the rows): what membership queries, k > 64 and indexes of more than 255
    genomes read (format 4 at 100 genomes, format 12 at 500).  Same results as on all the rows (row_source 9), conservation and
    membership, every kernel family; info.last_rows_read says what was read."""
    from memo_amd import synth
    L = 1_500_000
    ix, (r0, r1) = synth.device_index(0, L, 256, n, L, pack="only")
    num, den = synth.rows_per_position(n)
    with ix:
        assert ix.info()["packed_format"] == (4 if n <= 255 else 12)
        ix.set_option(3, 0)       # MEMO_OPT_BUILD_COST_PCT 0: the first query of a class builds its view
        for k, cap in ((4, 4), (7, 6), (9, 8), (17, 16), (21, 20), (31, 30), (33, 32), (34, 40), (50, 56), (64, 64), (65, 64), (101, 112),
                       (129, 128), (130, None)):
            for memb in (False, True):
                qs, qe = (40_000, 40_000 + 60_000) if memb else (4, L - 3)
                ix.debug_set_tuning(0, 0, 0, 9, 0)
                ref = ix.membership(qs, qe, k, n) if memb else ix.conservation(qs, qe, k, n)
                assert ix.info()["last_rows_read"] == r1 - r0
                ix.debug_set_tuning(0, 0, 0, 0, 0)
                for _ in range(2):
                    got = ix.membership(qs, qe, k, n) if memb else ix.conservation(qs, qe, k, n)
                    assert np.array_equal(got, ref), (k, memb)
                read = ix.info()["last_rows_read"]
                if cap is None or cap >= 56:                        # overlaps are uniform in 0 .. 59: caps of 60 and more spare nothing,
                                                                    # 56 less than the fifth a view has to spare
                    assert read == r1 - r0, (k, cap, read)
                else:
                    assert abs(read / (r1 - r0) - cap / 60) < 0.01, (k, cap, read)
        a, b = 700_000, 900_000
        sr0, sr1 = synth.shard_rows(a, b, 31, num, den, L)
        s, e, o = oracle.synth_rows(sr0, sr1 - sr0, num, den, n)
        assert np.array_equal(ix.conservation(a, b, 31, n), oracle.conservation(s, e, o, a, b, 31, n, literal=False))
        assert np.array_equal(ix.membership(a, a + 50_000, 31, n), oracle.membership(s, e, o, a, a + 50_000, 31, n, literal=False))
        assert ix.info()["device_bytes"] > 4 * (r1 - r0) * (1 + 8 / 60)      # the views are counted


def test_queries_of_one_index_on_several_streams(memo, oracle):
    """include/memo_amd.h, "Threads and streams": one thread may enqueue queries of one index on several streams.  What a
    query builds for later ones -- a tile table per (rows, tile width, k), a k-class view once it has become worth its pass -- is
    built on the stream of the query that needed it and read by the next query on ANOTHER stream, so it has to be complete
    when the call returns; an evicted tile table (more than four (k, view) pairs) must outlive the sweeps queued on it.
    Thirty-six queries, six k, three streams, nothing synchronised until the end: every result equals the oracle's."""
    import torch
    from memo_amd import synth
    n, L = 100, 600_000
    ix, (r0, r1) = synth.device_index(0, L, 256, n, L, pack="dense")
    num, den = synth.rows_per_position(n)
    s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, n)
    streams = [torch.cuda.Stream(device=0) for _ in range(3)]
    ks = (21, 31, 17, 48, 64, 9)
    qs, qe = 8, L - 5
    want = {k: oracle.conservation(s, e, o, qs, qe, k, n, literal=False).astype(np.uint8) for k in ks}
    outs = []
    with ix:
        for i in range(36):
            k, st = ks[i % 6], streams[(i // 2) % 3]
            out = torch.empty(qe - qs, dtype=torch.uint8, device="cuda:0")
            ix.conservation_u8_dev(qs, qe, k, n, out, stream=st.cuda_stream)
            outs.append((k, out))
        torch.cuda.synchronize()
        ix.check()
        assert ix.info()["last_sweep"] == 5
        for i, (k, out) in enumerate(outs):
            assert np.array_equal(out.cpu().numpy(), want[k]), (i, k)


@pytest.mark.parametrize("n_ranks,workload,extra", [(2, "c3", []), (3, "c2", []), (2, "c2", ["--plain-gather"]),
                                                     (2, "c4", []), (2, "c3", ["--root-weight", "0.3"]),
                                                     (2, "c5", []), (2, "c5", ["--k", "21"]), (2, "c5", ["--plain-gather"]),
                                                     (2, "c5", ["--k", "101", "--coding", "runs"]),
                                                     (8, "c5", ["--k", "21"]), (8, "c3", [])])
def test_bench_two_ranks_on_one_gpu(n_ranks, workload, extra):
    """`python bench.py --gpus N` as the driver spells it, with N > 1 RANKS for the first time on this pool's one-GPU boxes:
    RCCL refuses two ranks on a device, so the ranks share GPU 0 (MEMO_BENCH_ONE_DEVICE=1) and gloo carries the bytes through
    host memory (MEMO_BENCH_BACKEND=gloo; bench.py marks the line "test_transport").  Everything else is the N > 1 path as
    an 8-GPU node runs it: self-launch through torch.distributed.run, link probe, coding choice from the ranks' statistics,
    root weight, double-buffered send / receive per step, decode on rank 0, the gathered slice of the last rank against the
    oracle, every slice complete, one JSON line.  Round 5: the same with EIGHT ranks -- what the first contact with an 8-GPU node runs:
    seven receives and ONE decode launch for seven slices per step (memo_transport_runs_unpack_many_dev), a root weight chosen for
    seven peers; eight config-5 shards on the dense rows share the one GPU (they build their indexes one after the other)."""
    import json
    import subprocess
    import sys
    steps, warm = ("2", "1") if n_ranks > 3 else ("4", "2")
    env = dict(os.environ, MEMO_BENCH_ONE_DEVICE="1", MEMO_BENCH_BACKEND="gloo", MEMO_BENCH_ASSUME_DEVICES=str(n_ranks))
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n_ranks), "--steps", steps, "--warmup", warm,
                        "--workload", workload] + extra, capture_output=True, text=True, env=env, timeout=1800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == n_ranks and j["steps"] == int(steps) and j["scaling"] == "weak" and "test_transport" in j
    assert j["ranks_seen"]["world_size"] == n_ranks and len(j["ranks_seen"]["ranks"]) == n_ranks
    assert j["gather_parity_sample"]["equal_to_oracle"] is True and j["gather_parity_sample"]["rank"] == n_ranks - 1
    assert j["gather_parity_sample"].get("every_slice_complete", True) is True
    assert j["link_GBs_measured"] and j["link_probe"]["bytes_per_rank"] > 0
    assert j["value"] > 0 and j["config"]["gather_payload"]
    if workload == "c5":      # BASELINE's multi-GPU config: uint16 slices -- coded by the runs coding unless plain was asked for
        assert j["config"]["result_bytes_per_position"] == 2
        if "--plain-gather" in extra:
            assert j["config"]["gather_payload"].startswith("plain result bytes (67108864 B")
        else:
            cands = j["config"]["gather_coding_choice"]["candidates"]
            assert set(cands) == {"plain", "runs"} and cands["runs"]["wire_bytes"] < 0.3 * cands["plain"]["wire_bytes"]
            if "--coding" in extra:
                assert j["config"]["gather_coding_choice"]["picked"] == "runs" and "2 byte(s) per change" in j["config"]["gather_payload"]
    # round 6: next to every measured N > 1 value, what the step model makes of this run's own timings; and next to a headline that
    # is not config 5, a short config-5 leg -- the configuration BASELINE's multi-GPU target is set on -- with its own parity sample
    exp = j["expected_from_model"]
    assert exp["value"] > 0 and exp["step_ms"] > 0 and exp["bound_by"] in ("peer_sweep_plus_encode_ms", "root_sweep_plus_decode_ms", "link_ms")
    if workload == "c4":
        assert "membership slices have no transport coding" in j["config"]["gather_payload"]
    if workload != "c5":
        c5 = j["config5"]
        assert "error" not in c5, c5
        assert c5["value"] > 0 and c5["x_one_gpu"] > 0 and c5["x_one_gpu_model"] > 0 and c5["coding"] in ("runs", "plain")
        assert c5["gather_parity_sample"]["equal_to_oracle"] is True and c5["gather_parity_sample"].get("every_slice_complete", True) is True
        assert c5["expected_from_model"]["coding"] == c5["coding"]
    else:
        assert "config5" not in j
    if n_ranks == 8:          # all seven peers' slices were received and decoded, and the choice was made for seven peers
        choice = j["config"]["gather_coding_choice"]
        assert choice["candidates"]["runs"]["decode_ms_all_slices_one_launch"] is not None
        assert choice["picked"] == "runs" or workload == "c3", choice["picked"]
        assert 0 < choice["root_weight"] <= 1 and len({r["rank"] for r in j["ranks_seen"]["ranks"]}) == 8


def test_no_room_on_the_device_for_views_and_tile_tables(memo, oracle, ab):
    """Views and tile tables are optimisations: when the device has no memory left for them (memo_debug_fail_side_allocations
    of the AB library makes those allocations fail the way a full device does) queries answer from the rows they have -- same results, no error,
    info.last_variant / last_rows_read say that no table and no view were used.  Row dropping at pack time keeps every row."""
    from memo_amd import synth
    n, L = 100, 800_000
    num, den = synth.rows_per_position(n)
    ab.check(ab.lib().memo_debug_fail_side_allocations(1))
    try:
      for pack in ("dense", "only"):
        ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack=pack)
        s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, n)
        with ix:
            ix.set_option(3, 0)                                # (every query tries to build its view, and finds no room)
            for k in (21, 31, 64):
                want = oracle.conservation(s, e, o, 4, L - 3, k, n, literal=False)
                for _ in range(7):
                    assert np.array_equal(ix.conservation(4, L - 3, k, n), want), (pack, k)
                inf = ix.info()
                assert inf["last_rows_read"] == r1 - r0 and inf["last_variant"] == 0, (pack, k, inf)
                assert inf["views_resident"] == 0 and inf["tile_tables_resident"] == 0 and inf["side_bytes"] == 0
            wantm = oracle.membership(s, e, o, 1000, 60_000, 31, n, literal=False)
            for _ in range(7):
                assert np.array_equal(ix.membership(1000, 60_000, 31, n), wantm)
            assert ix.prepare(31, n) == 0                     # (an explicit prepare finds no room either: no error, nothing taken)
    finally:
        ab.check(ab.lib().memo_debug_fail_side_allocations(0))
    ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack="dense")
    with ix:
        ix.set_option(3, 0)
        ix.set_option(4, 5)
        for _ in range(2):
            ix.conservation(4, L - 3, 21, n)
        inf = ix.info()
        assert inf["last_variant"] == 2 and abs(inf["last_rows_read"] / (r1 - r0) - 20 / 60) < 0.01      # (table + view of overlaps < 20)


def test_views_stay_within_their_budget(memo, oracle, ab):
    """All the k-class views of one row source together stay within MEMO_OPT_VIEW_BUDGET_PCT (200 %) of the bytes of the rows
    they are views of; past that the least recently used view is retired (with the tile tables made for it; freed by the next
    memo_query_check, no wait on the query path) and its class backs off: it is looked at again only after four times as many
    queries.  Every class of the dense rows and of the 4-byte words in turn, twice around: results equal the sweep of all the
    rows throughout, the index never grows past rows + 2 x rows (+ tables), the second turn rebuilds next to nothing, and a
    class that was evicted does come back when it is asked often enough."""
    from memo_amd import synth
    n, L = 100, 1_200_000
    for pack, row_bytes in (("dense", 3.2), ("only", 4.0)):
        ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack=pack)
        with ix:
            ix.set_option(3, 0)                     # MEMO_OPT_BUILD_COST_PCT 0: a class that is looked at builds its view
            ix.set_option(4, 5)
            base = ix.info()["device_bytes"]
            rows_bytes = row_bytes * (r1 - r0)
            builds = []
            for turn in range(2):
                for k in range(3, 34, 2):
                    ix.debug_set_tuning(0, 0, 0, 9, 0)
                    ref = ix.conservation(8, L - 4, k, n, dtype=np.uint8)
                    ix.debug_set_tuning(0, 0, 0, 0, 0)
                    for _ in range(6):
                        assert np.array_equal(ix.conservation(8, L - 4, k, n, dtype=np.uint8), ref), (pack, turn, k)
                    inf = ix.info()
                    assert inf["device_bytes"] <= base + 2.0 * rows_bytes + 64 * (L // 32 + 64) * 8 + (1 << 20), (pack, turn, k, inf["device_bytes"], base)
                builds.append(ix.info()["view_builds"])
            assert builds[0] >= 12, (pack, builds)            # (k - 1 <= 48 of 60 spares a fifth: every class here builds its view)
            assert builds[1] - builds[0] <= 2, (pack, builds)  # back-off: six queries do not bring an evicted class back
            # ... sixteen more of them do (the class evicted first: k = 3)
            ix.debug_set_tuning(0, 0, 0, 9, 0)
            ref = ix.conservation(8, L - 4, 3, n, dtype=np.uint8)
            ix.debug_set_tuning(0, 0, 0, 0, 0)
            for _ in range(24):
                assert np.array_equal(ix.conservation(8, L - 4, 3, n, dtype=np.uint8), ref)
            inf = ix.info()
            assert inf["view_builds"] == builds[1] + 1 and inf["last_rows_read"] < r1 - r0, (pack, builds, inf)


def test_index_info_is_versioned(memo):
    """memo_index_info_t carries its size: the caller sets struct_bytes, the library writes no more than that (whole leading
    fields) and says how much it wrote -- a binder built against an older, shorter struct keeps its stack"""
    import ctypes as C
    from memo_amd import synth, _lib
    ix, _ = synth.device_index(0, 50_000, 31, 10, 50_000, pack="only")
    with ix:
        full = ix.info()
        assert full["struct_bytes"] == C.sizeof(_lib.IndexInfo) and full["version"] == 5 and full["rows"] > 0
        starts = sorted({getattr(_lib.IndexInfo, f).offset for f, _ in _lib.IndexInfo._fields_} | {C.sizeof(_lib.IndexInfo)})
        for cut in (16, 20, 24, 30, 88, 136, 141, C.sizeof(_lib.IndexInfo) - 8, C.sizeof(_lib.IndexInfo), C.sizeof(_lib.IndexInfo) + 64):
            buf = (C.c_ubyte * (C.sizeof(_lib.IndexInfo) + 128))(*([0xA5] * (C.sizeof(_lib.IndexInfo) + 128)))
            C.cast(buf, C.POINTER(C.c_uint32))[0] = cut
            _lib.check(_lib.lib().memo_index_get_info_v5(ix._h, C.cast(buf, C.POINTER(_lib.IndexInfo))))
            wrote = C.cast(buf, C.POINTER(C.c_uint32))[0]
            assert wrote == max(s for s in starts if s <= cut)                    # (a size inside a field: down to the field's start)
            assert all(b == 0xA5 for b in bytes(buf)[wrote:]), cut          # the guard bytes behind the caller's struct
            got = _lib.IndexInfo.from_buffer_copy(bytes(buf)[:wrote] + bytes(C.sizeof(_lib.IndexInfo) - wrote))
            assert got.version == 5 and got.rows == full["rows"]
            if cut >= 136:
                assert got.last_rows_read == full["last_rows_read"] and got.max_annot == full["max_annot"]
        for bad in (0, 8, 15):
            inf = _lib.IndexInfo()
            inf.struct_bytes = bad
            assert _lib.lib().memo_index_get_info_v5(ix._h, C.byref(inf)) == _lib.MEMO_EINVAL
            assert b"struct_bytes" in _lib.lib().memo_last_error()


def test_prepare_builds_views_and_tables_now(memo, oracle):
    """memo_index_prepare: the k-class view, the tile table and (builder-made indexes) the query order of the rows, built
    before the first query instead of by the query that finds them worth their pass; idempotent; nothing is launched into the caller's buffer"""
    from memo_amd import synth
    n, L = 100, 2_000_000
    num, den = synth.rows_per_position(n)
    for pack in ("dense", "only"):
        ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack=pack)
        s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, n)
        with ix:
            ix.set_option(4, 5)
            before = ix.info()
            taken = ix.prepare(31, n)
            inf = ix.info()
            assert taken > 0 and inf["device_bytes"] == before["device_bytes"] + taken and inf["side_bytes"] == taken
            assert inf["views_resident"] == 1 and inf["view_builds"] == 1 and inf["last_view_ms"] > 0
            assert inf["tile_tables_resident"] == (1 if pack == "dense" else 0)
            assert ix.prepare(31, n) == 0 and ix.info()["view_builds"] == 1       # everything is there
            got = ix.conservation(0, L, 31, n, dtype=np.uint8)                  # the FIRST query reads the view
            inf = ix.info()
            assert abs(inf["last_rows_read"] / (r1 - r0) - 0.5) < 0.01 and inf["view_builds"] == 1
            if pack == "dense":
                assert inf["last_sweep"] == 5 and inf["last_variant"] == 2
            assert np.array_equal(got, oracle.conservation(s, e, o, 0, L, 31, n, literal=False))
            if pack == "only":                                                  # membership of the same class: the same view
                assert ix.prepare(31, n, membership=True) == 0
                m = ix.membership(1000, 400_000, 31, n)
                assert ix.info()["last_rows_read"] == inf["last_rows_read"]
                assert np.array_equal(m, oracle.membership(s, e, o, 1000, 400_000, 31, n, literal=False))
            if pack == "only":
                assert ix.prepare(101, n) == 0                                  # every row can write at k = 101: no view pays
            ix.set_option(1, 0)                                                 # MEMO_OPT_VIEWS off: the views go
            inf = ix.info()
            assert inf["views_resident"] == 0 and inf["device_bytes"] <= before["device_bytes"] + (1 << 20)
            assert np.array_equal(ix.conservation(0, L, 31, n, dtype=np.uint8), got) and ix.info()["last_rows_read"] == r1 - r0
            ix.set_option(1, 1)
    # an index that came in through the builder: start order until prepare
    s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, n)
    with memo.DeviceIndex.from_host_packed(s, e, o) as ix:
        assert ix.info()["row_order"] == 0
        ix.prepare(101, n)
        assert ix.info()["row_order"] == 2
        assert np.array_equal(ix.conservation(0, L, 101, n), oracle.conservation(s, e, o, 0, L, 101, n, literal=False))
    with memo.DeviceIndex.from_host_packed(s, e, o) as ix:   # ... or until the queries have lost to the start order what the pass costs
        want = oracle.conservation(s, e, o, 0, L, 101, n, literal=False)
        for i in range(6):                                     # (six queries of 2 * 10^6 positions have not: 1.7 us against 340)
            assert np.array_equal(ix.conservation(0, L, 101, n), want)
            assert ix.info()["row_order"] == 0, i
        ix.set_option(3, 0)                                    # MEMO_OPT_BUILD_COST_PCT 0: now
        assert np.array_equal(ix.conservation(0, L, 101, n), want) and ix.info()["row_order"] == 2
        ix.check()
        assert np.array_equal(ix.conservation(5, L - 9, 101, n), want[5:L - 9])
    # WHICH order follows from the kind of query that pays for the pass: membership deals a bucket's rows over annot mod 32
    # (interleave mode 3), conservation over the bucket's starts (mode 2); rows in the other kind's order are re-ordered by the same rule
    with memo.DeviceIndex.from_host_packed(s, e, o) as ix:
        wantm = oracle.membership(s, e, o, 1000, 300_000, 31, n, literal=False)
        ix.prepare(31, n, membership=True)
        assert ix.info()["row_order"] == 3
        assert np.array_equal(ix.membership(1000, 300_000, 31, n), wantm)
        assert np.array_equal(ix.conservation(0, L, 101, n), want) and ix.info()["row_order"] == 3      # one query has not paid for a pass
        ix.set_option(3, 0)                                                                              # ... now it has
        assert np.array_equal(ix.conservation(0, L, 101, n), want) and ix.info()["row_order"] == 2
        assert np.array_equal(ix.membership(1000, 300_000, 31, n), wantm) and ix.info()["row_order"] == 3
        ix.check()
    with memo.DeviceIndex.from_host(s, e, o) as ix:
        ix.pack()
        assert ix.info()["row_order"] == 2                      # (memo_index_pack cannot know: the conservation order)
        for _ in range(3):
            assert np.array_equal(ix.membership(1000, 300_000, 31, n), wantm)
        assert ix.info()["row_order"] == 2                      # (three queries of 3 * 10^5 positions have not paid for the pass)
        ix.prepare(31, n, membership=True)
        assert ix.info()["row_order"] == 3
        assert np.array_equal(ix.membership(1000, 300_000, 31, n), wantm)
        ix.check()


def test_views_are_built_when_they_have_paid_for_themselves(memo, oracle):
    """no fixed "fifth query".  Every query of a k class that runs without its view adds what the view would
    have saved it (the rows of its window the view leaves out x what a sweep pays per row); the view is built by the query that finds
    the sum has reached the pass's estimated cost (ski rental: MEMO_OPT_BUILD_COST_PCT 100).  On a 10^7-position index of 5 * 10^7 rows
    (k = 31: a view spares half of them, 12 us per whole-chromosome query, against ~0.4 ms for the pass): ten whole-chromosome queries
    build nothing, two hundred build one view, ten thousand windows of 10^4 positions build nothing; memo_index_prepare builds at
    once; the same for the views of the 4-byte words."""
    import ctypes as C
    from memo_amd import synth, _lib
    n, L, k = 100, 10_000_000, 31
    num, den = synth.rows_per_position(n)
    d = C.c_void_p()
    _lib.check(_lib.lib().memo_dev_malloc(0, L, C.byref(d)))
    try:
        for pack in ("dense", "only"):
            sr0, sr1 = synth.shard_rows(2_000_000, 2_300_000, k, num, den, L)
            s, e, o = oracle.synth_rows(sr0, sr1 - sr0, num, den, n)
            want = oracle.conservation(s, e, o, 2_000_000, 2_300_000, k, n, literal=False).astype(np.uint8)

            def queries(ix, count, qs, qe):
                for _ in range(count):
                    ix.conservation_u8_dev(qs, qe, k, n, d.value)
                ix.check()

            ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack=pack)
            with ix:
                queries(ix, 10, 0, L)
                inf = ix.info()
                assert inf["view_builds"] == 0 and inf["last_rows_read"] == r1 - r0, (pack, inf)
                queries(ix, 190, 0, L)
                inf = ix.info()
                assert inf["view_builds"] == 1 and inf["views_resident"] == 1 and inf["last_rows_read"] < 0.55 * (r1 - r0), (pack, inf)
                assert np.array_equal(ix.conservation(2_000_000, 2_300_000, k, n, dtype=np.uint8), want)
            ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack=pack)
            with ix:
                rng = np.random.default_rng(5)
                for a in rng.integers(0, L - 10_000, 10_000):
                    ix.conservation_u8_dev(int(a), int(a) + 10_000, k, n, d.value)
                ix.check()
                assert ix.info()["view_builds"] == 0, pack
                assert ix.prepare(k, n) > 0 and ix.info()["view_builds"] == 1           # asked for: now
                assert np.array_equal(ix.conservation(2_000_000, 2_300_000, k, n, dtype=np.uint8), want)
                assert ix.info()["last_rows_read"] < 0.55 * (r1 - r0)
            ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack=pack)
            with ix:                                                                  # a lazier and an eager host
                ix.set_option(3, 1000)                                                # MEMO_OPT_BUILD_COST_PCT
                queries(ix, 200, 0, L)
                assert ix.info()["view_builds"] == 0, pack
                ix.set_option(3, 0)
                queries(ix, 1, 0, L)
                assert ix.info()["view_builds"] == 1, pack
    finally:
        _lib.lib().memo_dev_free(0, d)


def test_cycling_through_k_classes_does_not_thrash(memo, oracle):
    """a long-lived index queried across more k classes than round 3's four tile-table slots (and more than the
    view budget holds).  After the warm-up nothing is rebuilt: every class finds its view and its tile table again, round
    after round; with a budget that cannot hold them all the library settles instead of rebuilding a view every few queries."""
    from memo_amd import synth
    n, L = 100, 1_500_000
    num, den = synth.rows_per_position(n)
    ks = (5, 9, 13, 17, 21, 25, 29, 33)                     # eight classes of the dense rows
    ix, (r0, r1) = synth.device_index(0, L, 64, n, L, pack="dense")
    s, e, o = oracle.synth_rows(r0, r1 - r0, num, den, n)
    want = {k: oracle.conservation(s, e, o, 0, L, k, n, literal=False) for k in ks}
    with ix:
        ix.set_option(2, 800)                               # MEMO_OPT_VIEW_BUDGET_PCT: room for all eight
        ix.set_option(4, 5)
        for k in ks:
            ix.prepare(k, n)
        warm = ix.info()
        assert warm["views_resident"] == 8 and warm["tile_tables_resident"] == 8 and warm["view_builds"] == 8
        for _ in range(5):
            for k in ks:
                assert np.array_equal(ix.conservation(0, L, k, n, dtype=np.uint8), want[k]), k
                inf = ix.info()
                assert inf["last_variant"] == 2 and inf["last_rows_read"] < r1 - r0
        inf = ix.info()
        assert (inf["view_builds"], inf["views_resident"], inf["tile_tables_resident"]) == (8, 8, 8), inf
        assert inf["device_bytes"] == warm["device_bytes"]
    ix, _ = synth.device_index(0, L, 64, n, L, pack="dense")
    with ix:                                                # the default budget (200 %) holds about five of these classes
        ix.set_option(3, 0)                                 # (a class that is looked at builds: the back-off alone keeps the peace)
        for rnd in range(12):
            for k in ks:
                for _ in range(2):
                    assert np.array_equal(ix.conservation(0, L, k, n, dtype=np.uint8), want[k]), k
        inf = ix.info()
        assert inf["view_builds"] <= 16, inf                # (round 3's rule rebuilt an evicted class every fifth query: ~40 here)
        assert inf["side_bytes"] <= 2.1 * 3.2 * (r1 - r0) + 16 * (L // 32 + 64) * 8 + (8 << 20)


def test_row_order_inside_buckets_never_changes_a_result(memo, oracle, ab):
    """memo_interleave.hip: the 4-byte rows of a bucket dealt round-robin over the bucket's starts (chunks of four; the rows
    of a start by overlap mod 32).  Start order, both dealt orders and back again: every sweep family, conservation and
    membership, views included, on a ragged index with clumps (a bucket of more than 8192 rows stays as it is), both
    formats; equal to the oracle and to each other; the order is a function of the bucket's rows alone (idempotent)."""
    rng = np.random.default_rng(41)
    for n_docs, length, m in ((90, 60_000, 700_000), (600, 30_000, 900_000)):
        s = rng.integers(1, length, m)
        s[:9000] = 12_345                                    # a clump: > 8192 rows in one bucket
        s[9000:12000] = rng.integers(20_000, 20_032, 3000)   # a full bucket of 3000 rows
        s = np.sort(s).astype(np.int64)
        ov = rng.integers(0, 70, m)
        ov[::13] = rng.integers(200, 400, len(ov[::13]))
        e = s + ov
        o = rng.integers(1, n_docs, m).astype(np.int64)
        with memo.DeviceIndex.from_host(s, e, o) as ix:
            ix.set_option(3, 0)
            ix.debug_row_order(1)
            ix.pack(keep_wide=False)
            assert ix.info()["row_order"] == 0
            words0 = _export(ix)[0].copy()
            inf0 = ix.info()
            results = {}
            for order in (1, 2, 3, 2, 1):
                ix.debug_row_order(order)
                assert ix.info()["row_order"] == order - 1
                words = _export(ix)[0]
                assert np.array_equal(np.sort(words), np.sort(words0))          # a permutation of the same words ...
                boff = _export(ix)[2]
                for b in (0, len(boff) // 3, len(boff) // 2):                   # ... bucket by bucket
                    assert np.array_equal(np.sort(words[boff[b]:boff[b + 1]]), np.sort(words0[boff[b]:boff[b + 1]]))
                came_back = order == 1 and ("seen", 2) in results
                results[("seen", order)] = True
                if came_back:                                                   # back in start order (the rows of a start by overlap, annot)
                    f12 = inf0["packed_format"] == 12
                    st = ((words >> 8) & 0xFFF) if f12 else (words & 0xFFFF)
                    ovl = (words & 0xFF) if f12 else ((words >> 16) & 0xFF)
                    ann = (words >> 20) if f12 else (words >> 24)
                    key = ((st.astype(np.int64) & 31) << 20) | (ovl.astype(np.int64) << 12) | ann.astype(np.int64)
                    for b in (0, 5, len(boff) // 3, len(boff) // 2, 625):
                        assert boff[b + 1] - boff[b] > 8192 or np.all(np.diff(key[boff[b]:boff[b + 1]]) >= 0), b
                again = words.copy()
                ix.debug_row_order(order)                                       # idempotent
                assert np.array_equal(_export(ix)[0], again)
                for k in (3, 21, 31, 64, 101, 200, 256):
                    for rep in range(2 if k in (21, 101) else 1):               # (MEMO_OPT_BUILD_COST_PCT 0: the first query of a class builds its view)
                        qs, qe = 16, length + 40
                        got = ix.conservation(qs, qe, k, n_docs)
                    want = results.setdefault((k, "c"), oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs,
                                                                            literal=False))
                    assert np.array_equal(got, want), (n_docs, order, k)
                    gm = ix.membership(9_000, 24_000, k, n_docs)
                    wm = results.setdefault((k, "m"), oracle.membership(*oracle.filter_rows(s, e, o, 9_000, 24_000, k), 9_000, 24_000, k,
                                                                        n_docs, literal=False))
                    assert np.array_equal(gm, wm), (n_docs, order, k)


def test_places_inside_a_dense_group_never_change_a_result(memo, oracle, ab):
    """memo_view.hip, view_place_bucket: the rows of a dense k-class view get their place inside their 16-byte group (and
    their group inside the bucket) chosen against LDS bank conflicts.  Views built with and without it, on a ragged index
    with empty stretches, a few buckets above the kernel's 128-row limit and windows that begin inside a bucket: the same rows
    read, results equal to the oracle and to each other, every k of every class of two."""
    rng = np.random.default_rng(43)
    n_docs, length, m = 120, 90_000, 260_000
    s = rng.integers(1, length, m)
    s[:500] = rng.integers(30_016, 30_048, 500)          # one bucket of > 128 rows in the views: stays as it is
    s[500:700] = rng.integers(50_000, 50_032, 200)
    s[(s > 70_000) & (s < 72_000)] = 69_999              # an empty stretch
    s = np.sort(s).astype(np.int64)
    e = s + rng.integers(0, 64, m)
    o = rng.integers(1, n_docs, m).astype(np.int64)
    results = {}
    try:
        for colour in (1, 0, 1):
            ab.check(ab.lib().memo_debug_view_colouring(colour))
            with memo.DeviceIndex.from_host(s, e, o) as ix:
                ix.pack(keep_wide=False)
                ix.pack_dense(keep_packed=False)
                ix.set_option(4, 5)                                              # (five rows per group; six: test_six_row_views)
                for k in (2, 3, 8, 9, 16, 17, 21, 30, 31, 32, 33):
                    ix.prepare(k, n_docs)
                    for qs, qe in ((0, length + 50), (30_001, 50_017), (69_990, 72_100)):
                        got = ix.conservation(qs, qe, k, n_docs)
                        inf = ix.info()
                        assert inf["last_sweep"] == 5, (k, inf["last_sweep"])
                        key = (k, qs, qe)
                        if key not in results:
                            results[key] = (oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False),
                                            inf["last_rows_read"])
                        assert np.array_equal(got, results[key][0]), (colour, key)
                        assert inf["last_rows_read"] == results[key][1], (colour, key)
                    assert inf["last_rows_read"] < m                              # (a view was read)
                ix.check()
    finally:
        ab.check(ab.lib().memo_debug_view_colouring(1))


def _dense_rows_fields(groups, rows):
    """(B, A) of the first `rows` rows of dense five-row groups (memo_amd/csrc/memo_sweep.h: PackedRows3)"""
    g = groups.reshape(-1, 4).astype(np.uint64)
    hi = (g[:, 3] >> 16) & 0x1F
    B = np.empty((len(g), 5), np.uint64)
    A = np.empty((len(g), 5), np.uint64)
    for j in range(4):
        B[:, j] = g[:, j] & 0xFFFF
        A[:, j] = (g[:, j] >> 24) | (((hi >> j) & 1) << 8)
    B[:, 4] = ((g[:, 0] >> 16) & 0xFF) | (((g[:, 1] >> 16) & 0xFF) << 8)
    A[:, 4] = ((g[:, 2] >> 16) & 0xFF) | (((hi >> 4) & 1) << 8)
    return B.reshape(-1)[:rows], A.reshape(-1)[:rows]


def _view_twin(dense, rows3, cap, rpg):
    """what a k-class view of exported dense rows has to be when its rows keep the order they come in -- NumPy; (groups, table,
    rows): five rows per group back to back with the kept-rows table, or six per group, every bucket padded to whole groups with
    copies of its last row, the table in units of places"""
    groups, table = dense
    B, A = _dense_rows_fields(groups, rows3)
    keep = (B & 63) < cap
    before = np.concatenate([np.zeros(1, np.int64), np.cumsum(keep, dtype=np.int64)])
    tab = before[table]
    Bv, Av = B[keep], A[keep]
    rows_v = len(Bv)
    if rpg == 6:
        n = np.diff(tab)
        ng = (n + 5) // 6
        tab6 = 6 * np.concatenate([np.zeros(1, np.int64), np.cumsum(ng)])
        slot_row = np.empty(tab6[-1], np.int64)
        for b in np.nonzero(n)[0]:
            q = np.arange(6 * ng[b])
            slot_row[tab6[b]:tab6[b + 1]] = tab[b] + np.minimum(q, n[b] - 1)
        Bs, As = Bv[slot_row], Av[slot_row]
        lo = ((Bs >> 6) & 31) | (np.minimum(Bs & 63, 31) << 5)
        bucket = (Bs >> 11) & 31
        lo, an, bucket = lo.reshape(-1, 6), (As & 0xFF).reshape(-1, 6), bucket.reshape(-1, 6)
        out = np.empty((len(lo), 4), np.uint64)
        out[:, 0] = lo[:, 0] | (lo[:, 4] << 10) | (an[:, 0] << 24)
        out[:, 1] = lo[:, 1] | (an[:, 4] << 10) | (an[:, 1] << 24)
        out[:, 2] = lo[:, 2] | (an[:, 5] << 10) | (bucket[:, 0] << 18) | (an[:, 2] << 24)
        out[:, 3] = lo[:, 3] | (lo[:, 5] << 10) | (an[:, 3] << 24)
        return out.astype(np.uint32).reshape(-1), tab6, rows_v
    pad = (-rows_v) % 5
    Bp = np.concatenate([Bv, np.zeros(pad, np.uint64)]).reshape(-1, 5)
    Ap = np.concatenate([Av, np.zeros(pad, np.uint64)]).reshape(-1, 5)
    hi = sum(((Ap[:, j] >> 8) & 1) << j for j in range(5))
    out = np.empty((len(Bp), 4), np.uint64)
    out[:, 0] = Bp[:, 0] | ((Bp[:, 4] & 255) << 16) | ((Ap[:, 0] & 255) << 24)
    out[:, 1] = Bp[:, 1] | ((Bp[:, 4] >> 8) << 16) | ((Ap[:, 1] & 255) << 24)
    out[:, 2] = Bp[:, 2] | ((Ap[:, 4] & 255) << 16) | ((Ap[:, 2] & 255) << 24)
    out[:, 3] = Bp[:, 3] | (hi << 16) | ((Ap[:, 3] & 255) << 24)
    return out.astype(np.uint32).reshape(-1), tab, rows_v


def _lds_cycles(rows_w, km1, rpg):
    """tools/view_order_model.py's cost of a view's rows (one integer per place, _view_rows): LDS cycles per row instruction and
    half-wave -- 32 consecutive groups, place i -- for the first and the second block of a row's interval"""
    w = rows_w[:len(rows_w) // (32 * rpg) * (32 * rpg)]
    if rpg == 5:
        s, ov = (w >> 6) & 1023, w & 63
    else:
        s, ov = (w & 31) | (((w >> 18) & 31) << 5), (w >> 5) & 31
    n = np.maximum(km1 - ov.astype(np.int64), 1)
    lev = np.floor(np.log2(n)).astype(np.int64)
    cells = [lev * 1024 + ((s.astype(np.int64) - n) & 1023), lev * 1024 + ((s.astype(np.int64) - (1 << lev)) & 1023)]
    total = 0.0
    for cell in cells:
        c = cell.reshape(-1, 32, rpg).transpose(0, 2, 1).reshape(-1, 32)      # one row per (half-wave, place)
        worst = np.zeros(len(c), np.int64)
        for b in range(32):
            on = (c & 31) == b
            srt = np.sort(np.where(on, c, -1), axis=1)
            same = (srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] >= 0)
            worst = np.maximum(worst, on.sum(1) + same.sum(1))               # a lane per different address, two per repeated one (- 1)
        total += np.maximum(worst, 2).mean()
    return total


def test_dense_views_as_the_fused_pass_builds_them(memo, oracle, ab):
    """memo_view.hip (round 5): a dense k-class view is built by count -> scan -> ONE fused pass (compaction, the places of the rows
    inside their groups, packing) instead of round 4's five kernels.  (1) Rows in the order they come: the exported bytes equal a NumPy
    twin's -- views of five rows per group and of six, the dense rows with the never-writing rows left out (dense_compact: the same pass),
    eight- and nine-bit annots, on a ragged index with empty stretches, buckets above the placing limits (96 / 128 rows) and one bucket
    whose kept rows do not fit the pass's LDS stage (it streams through in pieces).  (2) Rows placed: every bucket holds the same rows
    as before, and by the LDS cycle model the places are worth having (on an index of BASELINE's shape: a fifth fewer cycles).  (3) The
    sweep on every one of them equals the oracle."""
    rng = np.random.default_rng(67)
    length = 90_000
    for n_docs, m in ((120, 300_000), (500, 260_000)):
        s = rng.integers(1, length, m)
        s[:500] = rng.integers(30_016, 30_048, 500)          # > 128 rows of a view in one bucket
        s[500:700] = rng.integers(50_000, 50_032, 200)
        s[700:20_700] = rng.integers(60_000, 60_032, 20_000)  # more kept rows in one bucket than the fused pass stages (5120): pieces
        s[20_700:20_705] = 60_040                            # (five rows right behind it)
        s[(s > 70_000) & (s < 72_000)] = 69_999              # an empty stretch
        s = np.sort(s).astype(np.int64)
        ov = rng.integers(0, 64, m)
        ov[::7] = rng.integers(63, 90, len(ov[::7]))         # a seventh of the rows never writes at k <= 64: they leave the dense rows
        e = s + ov
        o = rng.integers(1, n_docs, m).astype(np.int64)
        try:
            for six in ((0, 1) if n_docs <= 255 else (0,)):
                ab.check(ab.lib().memo_debug_six_views(six))
                plain = {}
                for colour in (0, 1):
                    ab.check(ab.lib().memo_debug_view_colouring(colour))
                    with memo.DeviceIndex.from_host(s, e, o) as ix:
                        ix.pack(keep_wide=True)
                        pk = _export(ix)[0]
                        ix.pack(keep_wide=False)
                        ix.pack_dense(keep_packed=False)
                        inf = ix.info()
                        rows3 = inf["dense_row_count"]
                        assert rows3 < m                                      # (dense_compact ran: through the same pass)
                        dense = _export_dense(ix)
                        if colour == 0 and not six:                           # the dense rows themselves: the packed rows that can write at k <= 64
                            f12 = inf["packed_format"] == 12
                            st_, ov_, an_ = ((pk >> 8) & 0xFFF, pk & 0xFF, pk >> 20) if f12 else (pk & 0xFFFF, (pk >> 16) & 0xFF, pk >> 24)
                            stay = ov_ < 63
                            B, A = _dense_rows_fields(dense[0], rows3)
                            assert np.array_equal(B, (((st_[stay] & 1023) << 6) | ov_[stay]).astype(np.uint64)) and np.array_equal(A, an_[stay].astype(np.uint64))
                        for k in (3, 9, 17, 21, 31, 32, 33):
                            rpg = 6 if six and k <= 32 else 5
                            ix.prepare(k, n_docs)
                            view = ix.export_view(k, rpg)
                            assert view is not None, (colour, six, k)
                            twin = _view_twin(dense, rows3, view[3], rpg)
                            assert view[2] == twin[2] and np.array_equal(view[1], twin[1]), ("table", n_docs, colour, six, k)
                            if colour == 0:
                                same = view[0] == twin[0]
                                assert same.all(), ("groups", n_docs, six, k, int(np.argmin(same)) // 4, len(same) // 4)
                                plain[k] = view
                            else:
                                mine, theirs = _view_rows(view, rpg), _view_rows(plain[k], rpg)
                                assert len(mine) == len(theirs)
                                bucket_of = np.searchsorted(view[1][1:], np.arange(len(mine)), side="right")
                                assert np.array_equal(mine[np.lexsort((mine, bucket_of))], theirs[np.lexsort((theirs, bucket_of))]), (n_docs, six, k)
                            want = oracle.conservation(*oracle.filter_rows(s, e, o, 5, length + 50, k), 5, length + 50, k, n_docs, literal=False)
                            assert np.array_equal(ix.conservation(5, length + 50, k, n_docs), want), (colour, six, k)
                            assert ix.info()["last_view_placed"] == colour and ix.info()["last_view_rows_per_group"] == rpg
                        ix.check()
        finally:
            ab.check(ab.lib().memo_debug_view_colouring(1))
            ab.check(ab.lib().memo_debug_six_views(-1))
    # every bucket too long for the stage, runs of four buckets that begin inside a group (the fuzzer's find: 256-position buckets
    # of ~9000 rows): the dense rows and two views against the twin
    m, length, n_docs = 115_870, 3000, 33
    s = np.sort(rng.integers(60, 2900, m)).astype(np.int64)
    ov = rng.integers(0, 72, m)
    e = s + ov
    o = rng.integers(1, n_docs, m).astype(np.int64)
    with memo.DeviceIndex.from_host(s, e, o, bucket_shift=8) as ix:
        ix.pack(keep_wide=True)
        pk = _export(ix)[0]
        ix.pack_dense(keep_packed=False)
        inf = ix.info()
        assert inf["bucket_shift"] == 8 and inf["dense_row_count"] < m
        dense = _export_dense(ix)
        B, A = _dense_rows_fields(dense[0], inf["dense_row_count"])
        stay = ((pk >> 16) & 0xFF) < 63
        assert np.array_equal(B, ((((pk & 0xFFFF)[stay] & 1023) << 6) | ((pk >> 16) & 0xFF)[stay]).astype(np.uint64))
        assert np.array_equal(A, (pk >> 24)[stay].astype(np.uint64))
        for k in (5, 21):
            ix.prepare(k, n_docs)
            view = ix.export_view(k, 5)
            twin = _view_twin(dense, inf["dense_row_count"], view[3], 5)
            assert view[2] == twin[2] and np.array_equal(view[1], twin[1]) and np.array_equal(view[0], twin[0]), k
            want = oracle.conservation(*oracle.filter_rows(s, e, o, 1984, 2655, k), 1984, 2655, k, n_docs, literal=False)
            assert np.array_equal(ix.conservation(1984, 2655, k, n_docs), want), k
    # what the places are worth, by the cycle model, on BASELINE's shape (5 rows per position, overlaps uniform in 0 .. 59), k = 31
    from memo_amd import synth
    for rpg in (5, 6):
        cyc = {}
        for placed in (0, 1):
            ix, _ = synth.device_index(0, 600_000, 64, 100, 600_000, pack="dense")
            with ix:
                ix.set_option(4, rpg)
                ix.set_option(5, placed)
                ix.prepare(31, 100)
                view = ix.export_view(31, rpg)
                assert ix.info()["view_placings"] == placed
                cyc[placed] = _lds_cycles(_view_rows(view, rpg), 30, rpg)
        assert cyc[1] < 0.85 * cyc[0], (rpg, cyc)


def _view_rows(view, rpg):
    """the rows of an exported view as one integer per row (slot), in slot order: five-row groups: (B | annot << 16) of rows
    [0, rows); six-row groups: (lo | annot << 10 | bucket << 18) of every slot (the places a bucket leaves empty hold copies)"""
    g = view[0].reshape(-1, 4).astype(np.uint64)
    if rpg == 5:
        hi = (g[:, 3] >> 16) & 0x1F
        out = np.empty((len(g), 5), np.uint64)
        for j in range(4):
            out[:, j] = (g[:, j] & 0xFFFF) | (((g[:, j] >> 24) | (((hi >> j) & 1) << 8)) << 16)
        out[:, 4] = ((g[:, 0] >> 16) & 0xFF) | (((g[:, 1] >> 16) & 0xFF) << 8) | ((((g[:, 2] >> 16) & 0xFF) | (((hi >> 4) & 1) << 8)) << 16)
        return out.reshape(-1)[:view[2]]
    lo = [g[:, 0] & 0x3FF, g[:, 1] & 0x3FF, g[:, 2] & 0x3FF, g[:, 3] & 0x3FF, (g[:, 0] >> 10) & 0x3FF, (g[:, 3] >> 10) & 0x3FF]
    an = [g[:, 0] >> 24, g[:, 1] >> 24, g[:, 2] >> 24, g[:, 3] >> 24, (g[:, 1] >> 10) & 0xFF, (g[:, 2] >> 10) & 0xFF]
    bucket = (g[:, 2] >> 18) & 31
    return np.stack([lo[i] | (an[i] << 10) | (bucket << 18) for i in range(6)], axis=1).reshape(-1)


def _export_dense(ix):
    """(groups, bucket table) of the dense rows of an index"""
    from memo_amd import _lib
    inf = ix.info()
    groups = np.empty(4 * ((inf["dense_row_count"] + 4) // 5), np.uint32)
    boff = np.empty(inf["buckets"], np.int64)
    longs = np.empty(3 * inf["long_rows"], np.int64)
    _lib.check(_lib.lib().memo_index_export_dense(ix._h, groups.ctypes.data, boff.ctypes.data, longs.ctypes.data if longs.size else None))
    return groups, boff


def test_dense_rows_of_256_to_511_genomes(memo, oracle, ab):
    """Indexes of 256 .. 511 genomes on the dense rows (the ninth annot bit in the group's spare byte; the nine-bit forms of the
    table-driven kernel and of the one without a table: memo_sweep_cons3t.hip / memo_sweep_cons.hip, A9): conservation at every k class up to 64, views built and not, windows on and off the
    4-position raster, membership and k > 64 through the 4-byte rows (refused when those were dropped); an index whose annots would
    fit a byte asked with more than 255 genomes; equal to the oracle."""
    rng = np.random.default_rng(47)
    length = 70_000
    for n_docs, top_annot, m in ((500, 499, 300_000), (511, 510, 160_000), (300, 200, 160_000)):
        s = np.sort(rng.integers(1, length, m)).astype(np.int64)
        e = s + rng.integers(0, 70, m)
        o = rng.integers(1, top_annot + 1, m).astype(np.int64)
        o[:4] = (top_annot, 1, 255 if top_annot > 255 else 1, 256 if top_annot > 256 else 1)
        for keep_packed in (True, False):
            with memo.DeviceIndex.from_host(s, e, o) as ix:
                ix.pack(keep_wide=False)
                assert ix.info()["packed_format"] == (12 if top_annot > 255 else 4)
                ix.pack_dense(keep_packed=keep_packed)
                assert ix.info()["dense_rows"] == 1
                ix.set_option(3, 0)
                for k in (2, 3, 9, 21, 31, 32, 33, 48, 64):
                    if k in (21, 33):
                        ix.prepare(k, n_docs)
                    for rep in range(2 if k == 9 else 1):                       # (MEMO_OPT_BUILD_COST_PCT 0: the first query of a class builds its view)
                        got = ix.conservation(0, length + 60, k, n_docs)
                    inf = ix.info()
                    assert (inf["last_sweep"], inf["last_variant"]) == (5, 2), (n_docs, k, inf["last_sweep"], inf["last_variant"])
                    want = oracle.conservation(*oracle.filter_rows(s, e, o, 0, length + 60, k), 0, length + 60, k, n_docs, literal=False)
                    assert np.array_equal(got, want), (n_docs, keep_packed, k)
                    assert got.max() <= n_docs and (got == n_docs).any() == (want == n_docs).any()
                    for qs, qe in ((20_004, 41_003), (20_001, 41_003), (20_002, 20_009), (20_003, 41_000)):   # on and off the 4-position raster
                        got = ix.conservation(qs, qe, k, n_docs)
                        assert (ix.info()["last_sweep"], ix.info()["last_variant"]) == (5, 2)
                        assert np.array_equal(got, oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False))
                    # the kernel without a tile table (what a negative window start or a device without room for the table gets): nine bits too
                    for qs, qe, source in ((-70, 9_001, 0), (20_001, 41_003, 5), (0, length + 60, 5)):
                        ix.debug_set_tuning(0, 0, 0, source, 0)
                        got = ix.conservation(qs, qe, k, n_docs)
                        assert (ix.info()["last_sweep"], ix.info()["last_variant"]) == (5, 0), (k, qs, source)
                        assert np.array_equal(got, oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False))
                    ix.debug_set_tuning()
                if keep_packed:
                    for k in (31, 101):
                        got = ix.membership(9_000, 19_000, k, n_docs)
                        assert np.array_equal(got, oracle.membership(*oracle.filter_rows(s, e, o, 9_000, 19_000, k), 9_000, 19_000, k, n_docs,
                                                                     literal=False))
                    got = ix.conservation(0, length, 101, n_docs)
                    assert np.array_equal(got, oracle.conservation(*oracle.filter_rows(s, e, o, 0, length, 101), 0, length, 101, n_docs, literal=False))
                else:
                    with pytest.raises(memo.MemoError):
                        ix.membership(9_000, 19_000, 31, n_docs)
                ix.check()
    # 512 genomes and more: no dense rows
    o = rng.integers(1, 600, m).astype(np.int64)
    with memo.DeviceIndex.from_host(s, e, o) as ix:
        ix.pack(keep_wide=False)
        with pytest.raises(memo.MemoError, match="annot <= 511"):
            ix.pack_dense()


def test_six_row_views_equal_five_row_views(memo, oracle, ab):
    """The dense k-class views as groups of SIX rows that carry their bucket (memo_view.hip: view_build_kernel<6>; 2.67 B per row;
    the library's own choice where they apply since round 5) on the table-driven kernel's form for them (info.last_variant 3),
    forced on and off through memo_debug_six_views of the A/B library so that BOTH kinds answer every case.  Ragged index with an empty stretch and a bucket above the builder's 96-row limit, every k class it takes
    (k - 1 <= 31), windows that begin inside a bucket, both result types: equal to the oracle and to the five-row views."""
    rng = np.random.default_rng(61)
    n_docs, length, m = 120, 90_000, 260_000
    s = rng.integers(1, length, m)
    s[:500] = rng.integers(30_016, 30_048, 500)
    s[(s > 70_000) & (s < 72_000)] = 69_999
    s = np.sort(s).astype(np.int64)
    e = s + rng.integers(0, 64, m)
    o = rng.integers(1, n_docs, m).astype(np.int64)
    want = {}
    try:
        for six in (1, 0):
            ab.check(ab.lib().memo_debug_six_views(six))
            with memo.DeviceIndex.from_host(s, e, o) as ix:
                ix.pack(keep_wide=False)
                ix.pack_dense(keep_packed=False)
                for k in (2, 3, 9, 17, 21, 30, 31, 32, 33):
                    ix.prepare(k, n_docs)
                    for qs, qe in ((0, length + 50), (30_001, 50_017), (69_990, 72_100), (5, 7)):
                        for dt in (np.uint8, np.uint16):
                            got = ix.conservation(qs, qe, k, n_docs, dtype=dt)
                            inf = ix.info()
                            assert inf["last_sweep"] == 5 and inf["last_variant"] == (3 if six and k <= 32 else 2), (six, k, inf["last_variant"])
                            key = (k, qs, qe)
                            if key not in want:
                                want[key] = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                            assert np.array_equal(got, want[key].astype(dt)), (six, key, dt)
                ix.check()
    finally:
        ab.check(ab.lib().memo_debug_six_views(-1))


def test_level_arrays_follow_the_overlap_lengths(memo, oracle, ab):
    """k >= 65, unclipped conservation sweep: the library picks doubling / radix-4 / mixed level arrays from k and the
    overlap lengths it sampled when the packed rows were made (info.last_sweep says which ran); every choice is
    bit-exact"""
    rng = np.random.default_rng(77)
    n_docs, length, m = 60, 300_000, 1_500_000                     # 5 rows per position: "moderate" density
    s = np.sort(rng.integers(1, length, m)).astype(np.int64)
    o = rng.integers(1, n_docs, m).astype(np.int64)
    cases = {
        # (family, level arrays of the plan): mixed arrays wherever few intervals have fewer than 16 positions -- and of those
        # only the ones some row can write to (+ the blocks of 16 the fold goes through)
        "uniform 0..59": (rng.integers(0, 60, m), {31: 2, 64: 2, 65: 2, 80: (4, 3), 101: (4, 3), 128: (4, 2), 129: (4, 3), 160: (4, 3),
                                                   200: (4, 2), 256: (4, 2)}),
        "all 0": (np.zeros(m, np.int64), {65: (4, 2), 101: (4, 2), 200: (4, 2)}),   # n = k - 1: one length, one level
        "0..250": (rng.integers(0, 251, m), {65: 2, 101: 3, 200: 3, 256: 3}),  # many short intervals, every level populated: radix-4's
                                                                               # four arrays while a row takes few blocks, else doubling
    }
    for name, (ov, want_by_k) in cases.items():
        e = s + ov.astype(np.int64)
        for way in ("pack", "builder"):
            if way == "pack":
                ix = memo.DeviceIndex.from_host(s, e, o)
                ix.pack(keep_wide=False)
            else:
                ix = memo.DeviceIndex.from_host_packed(s, e, o)
            with ix:
                # (a window this short would get arrays so small that the k - 1 halo rules the large ones out:
                # keep the long-window shape; the choice of the levels stays the library's)
                ix.debug_set_tuning(tile_w=2560, waves=8)
                for k, want in want_by_k.items():
                    qs, qe = 1000, 250_000
                    got = ix.conservation(qs, qe, k, n_docs)
                    inf = ix.info()
                    want_fam, want_arrays = want if isinstance(want, tuple) else (want, None)
                    assert inf["last_sweep"] == want_fam, (name, way, k, inf["last_sweep"])
                    assert want_arrays is None or inf["last_level_arrays"] == want_arrays, (name, way, k, inf["last_level_arrays"])
                    ref = oracle.conservation(*oracle.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n_docs, literal=False)
                    assert np.array_equal(got, ref), (name, way, k)


def test_transport_nibble_coding_round_trip(memo):
    """uint8 results -> nibbles + exception list -> uint8, for value mixes with none, few and too
    many values >= 15"""
    import ctypes as C
    import torch
    from memo_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(6)
    for n, hi, frac in ((0, 10, 0), (1, 200, 1.0), (12345, 14, 0), (1_000_003, 200, 0.001), (1_000_000, 255, 0.01), (4096, 255, 1.0),
                        (300_000, 255, 0.1), (100_000, 255, 0.5)):
        v = rng.integers(0, 15, n).astype(np.uint8)
        big = rng.random(n) < frac
        v[big] = rng.integers(15, hi + 1, int(big.sum())) if hi >= 15 else v[big]
        cap = max(n // 64, 4)
        src = torch.from_numpy(v).cuda()
        wire = torch.zeros(L.memo_transport_bytes(n, cap), dtype=torch.uint8, device="cuda")
        dst = torch.full((max(n, 1),), 77, dtype=torch.uint8, device="cuda")
        _lib.check(L.memo_transport_pack_dev(src.data_ptr(), n, cap, wire.data_ptr(), 0, None))
        found, have = C.c_uint32(), C.c_uint32()
        _lib.check(L.memo_transport_exceptions(wire.data_ptr(), 0, None, C.byref(found), C.byref(have)))
        # a workgroup stages at most 4096 exceptions of its 32768 positions: beyond that the slice reports itself incomplete
        per_block = np.add.reduceat((v >= 15).astype(np.int64), np.arange(0, max(n, 1), 32768)) if n else np.zeros(1, np.int64)
        overflow = bool((per_block > 4096).any())
        assert found.value == (0xFFFFFFFF if overflow else int((v >= 15).sum())) and have.value == cap
        assert overflow == (frac == 0.5)
        _lib.check(L.memo_transport_unpack_dev(wire.data_ptr(), n, dst.data_ptr(), 0, None))
        torch.cuda.synchronize()
        if found.value <= cap:
            assert np.array_equal(dst[:n].cpu().numpy(), v), (n, hi, frac)
        assert wire.numel() == 16 + ((n + 7) // 8) * 4 + cap * 8


@pytest.mark.gpu
def test_transport_runs_coding_round_trip(memo, oracle):
    """uint8 results -> change bitmap + one byte per change (allocated exactly per 32768 positions) -> uint8:
    run-length-like data (what conservation looks like), data that changes at every position, constant data,
    ragged lengths around the 16-position pieces, 4096-position rounds and 32768-position blocks, a capacity that
    does not suffice (reported, never silent), and a real conservation result"""
    import ctypes as C
    import torch
    from memo_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(26)

    def round_trip(v, b_cap):
        n = len(v)
        src = torch.from_numpy(v).cuda() if n else torch.zeros(16, dtype=torch.uint8, device="cuda")
        wire = torch.zeros(L.memo_transport_runs_bytes(n, b_cap), dtype=torch.uint8, device="cuda")
        dst = torch.full((max(n, 16),), 77, dtype=torch.uint8, device="cuda")
        _lib.check(L.memo_transport_runs_pack_dev(src.data_ptr(), n, b_cap, wire.data_ptr(), 0, None))
        taken, room = C.c_uint32(), C.c_uint32()
        _lib.check(L.memo_transport_runs_stats(wire.data_ptr(), 0, None, C.byref(taken), C.byref(room)))
        _lib.check(L.memo_transport_runs_unpack_dev(wire.data_ptr(), n, b_cap, dst.data_ptr(), 0, None))
        torch.cuda.synchronize()
        return dst[:n].cpu().numpy(), taken.value, room.value, wire.numel()

    def wanted(v):
        n = len(v)
        ch = np.ones(n, bool)
        ch[1:] = v[1:] != v[:-1]
        ch[::32768] = True
        per_block = [int(ch[i:i + 32768].sum()) for i in range(0, n, 32768)]
        return sum((c + 3) & ~3 for c in per_block), len(per_block)

    for n in (0, 1, 15, 16, 17, 4095, 4096, 4097, 12345, 32767, 32768, 32769, 65536, 1_000_003, 3_000_000):
        for mix in ("runs", "uniform", "constant", "zeros", "alternating"):
            if mix == "runs":
                v = np.repeat(np.minimum(rng.geometric(0.33, n // 3 + 1), 255), rng.integers(1, 30, n // 3 + 1))[:n].astype(np.uint8)
                v = np.resize(v, n) if len(v) < n else v
            elif mix == "uniform":
                v = rng.integers(0, 256, n).astype(np.uint8)
            elif mix == "constant":
                v = np.full(n, 7, np.uint8)
            elif mix == "zeros":
                v = np.zeros(n, np.uint8)
            else:
                v = (np.arange(n) & 1).astype(np.uint8) * 255
            b_want, blocks = wanted(v)
            got, taken, room, size = round_trip(v, b_want)
            assert (taken, room) == (b_want, b_want), (n, mix, taken, room, b_want)
            assert np.array_equal(got, v), (n, mix)
            assert size == ((((16 + 8 * blocks + 4096 * blocks + 15) & ~15) + b_want + 15) & ~15), (n, mix)
            if mix == "runs" and n >= 1_000_000:
                got, taken, room, _ = round_trip(v, (b_want // 2) & ~3)          # too small: said so, never silent
                assert taken == b_want and room == (b_want // 2) & ~3 and taken > room
                got, taken, room, _ = round_trip(v, b_want + 4096)               # slack is fine
                assert taken == b_want and np.array_equal(got, v)
    # a conservation result: about one change in ten positions at k = 31
    from memo_amd import synth
    n_docs, length, k = 100, 2_000_000, 31
    ix, (r0, r1) = synth.device_index(0, length, k, n_docs, length, pack="only")
    with ix:
        v = ix.conservation(0, length, k, n_docs, dtype=np.uint8)
    b_want, blocks = wanted(v)
    got, taken, room, size = round_trip(v, b_want)
    assert np.array_equal(got, v) and taken == b_want
    assert size < 0.26 * length, size                                             # < 2.1 bits per position on the wire
    with pytest.raises(memo.MemoError):
        _lib.check(L.memo_transport_runs_pack_dev(1, 100, 6, 16, 0, None))       # capacity not a multiple of 4


def test_transport_runs_many_slices_one_launch(memo):
    """memo_transport_runs_unpack_many_dev: the slices of one gather step (as rank 0 receives them from its peers: same length,
    same capacity) decoded by ONE launch -- 1, 7 and 19 slices (more than one launch's 16), uint8 and uint16 values, lengths on
    and off the coding's 32768-position blocks; equal to decoding every slice by itself"""
    import ctypes as C
    import torch
    from memo_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(29)
    for vb, pack_fn, dt in ((1, L.memo_transport_runs_pack_dev, np.uint8), (2, L.memo_transport_runs16_pack_dev, np.uint16)):
        for n in (1, 4097, 32768, 100_001, 1_000_003):
            for count in (1, 7, 19):
                b_cap = (n * vb + 4 * (n // 32768 + 1) + 3) & ~3
                vals = [np.repeat(rng.integers(1, 60000 if vb == 2 else 250, n // 5 + 1), rng.integers(1, 10, n // 5 + 1))[:n] for _ in range(count)]
                vals = [np.resize(v, n).astype(dt) for v in vals]
                srcs = [torch.from_numpy(v.view(np.int16 if vb == 2 else np.uint8)).cuda() for v in vals]
                wires = [torch.zeros(L.memo_transport_runs_bytes(n, b_cap), dtype=torch.uint8, device="cuda") for _ in range(count)]
                outs = [torch.full((n + 16,), 7, dtype=torch.int16 if vb == 2 else torch.uint8, device="cuda") for _ in range(count)]
                for s_, w_ in zip(srcs, wires):
                    _lib.check(pack_fn(s_.data_ptr(), n, b_cap, w_.data_ptr(), 0, None))
                ws = (C.c_void_p * count)(*[w_.data_ptr() for w_ in wires])
                os_ = (C.c_void_p * count)(*[o_.data_ptr() for o_ in outs])
                _lib.check(L.memo_transport_runs_unpack_many_dev(ws, os_, count, n, b_cap, vb, 0, None))
                torch.cuda.synchronize()
                for i in range(count):
                    got = outs[i][:n].cpu().numpy().view(dt)
                    assert np.array_equal(got, vals[i]), (vb, n, count, i)
                    assert int(outs[i][n]) == 7                                      # nothing written behind a slice
    assert L.memo_transport_runs_unpack_many_dev(None, None, 0, 100, 400, 1, 0, None) == 0
    assert L.memo_transport_runs_unpack_many_dev(None, None, 2, 100, 400, 1, 0, None) == _lib.MEMO_EINVAL
    assert L.memo_transport_runs_unpack_many_dev(None, None, 0, 100, 400, 3, 0, None) == _lib.MEMO_EINVAL


def test_transport_runs16_coding_round_trip(memo, oracle):
    """the runs coding of uint16 results (more than 255 genomes: BASELINE config 5): change bitmap + TWO bytes per change;
    the same shapes as the uint8 test, values above 255, a capacity that does not suffice, and a 500-genome conservation
    result (~2.3 bits per position at k = 31 against 16 plain)"""
    import ctypes as C
    import torch
    from memo_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(27)

    def round_trip(v, b_cap):
        n = len(v)
        src = torch.from_numpy(v.view(np.int16)).cuda() if n else torch.zeros(16, dtype=torch.int16, device="cuda")
        wire = torch.zeros(L.memo_transport_runs_bytes(n, b_cap), dtype=torch.uint8, device="cuda")
        dst = torch.full((max(n, 16),), 77, dtype=torch.int16, device="cuda")
        _lib.check(L.memo_transport_runs16_pack_dev(src.data_ptr(), n, b_cap, wire.data_ptr(), 0, None))
        taken, room = C.c_uint32(), C.c_uint32()
        _lib.check(L.memo_transport_runs_stats(wire.data_ptr(), 0, None, C.byref(taken), C.byref(room)))
        _lib.check(L.memo_transport_runs16_unpack_dev(wire.data_ptr(), n, b_cap, dst.data_ptr(), 0, None))
        torch.cuda.synchronize()
        return dst[:n].cpu().numpy().view(np.uint16), taken.value, room.value, wire.numel()

    def wanted(v):
        n = len(v)
        ch = np.ones(n, bool)
        ch[1:] = v[1:] != v[:-1]
        ch[::32768] = True
        per_block = [int(ch[i:i + 32768].sum()) for i in range(0, n, 32768)]
        return sum((2 * c + 3) & ~3 for c in per_block), len(per_block)

    for n in (0, 1, 15, 16, 17, 4095, 4096, 4097, 12345, 32767, 32768, 32769, 65536, 1_000_003, 3_000_000):
        for mix in ("runs", "uniform", "constant", "alternating"):
            if mix == "runs":
                v = np.repeat(np.minimum(rng.geometric(0.01, n // 3 + 1), 65535), rng.integers(1, 30, n // 3 + 1))[:n].astype(np.uint16)
                v = np.resize(v, n) if len(v) < n else v
            elif mix == "uniform":
                v = rng.integers(0, 65536, n).astype(np.uint16)
            elif mix == "constant":
                v = np.full(n, 500, np.uint16)
            else:
                v = ((np.arange(n) & 1) * 65535).astype(np.uint16)
            b_want, blocks = wanted(v)
            got, taken, room, size = round_trip(v, b_want)
            assert (taken, room) == (b_want, b_want), (n, mix, taken, room, b_want)
            assert np.array_equal(got, v), (n, mix)
            assert size == ((((16 + 8 * blocks + 4096 * blocks + 15) & ~15) + b_want + 15) & ~15), (n, mix)
            if mix == "runs" and n >= 1_000_000:
                got, taken, room, _ = round_trip(v, (b_want // 2) & ~3)          # too small: said so, never silent
                assert taken == b_want and room == (b_want // 2) & ~3 and taken > room
                got, taken, room, _ = round_trip(v, b_want + 4096)               # slack is fine
                assert taken == b_want and np.array_equal(got, v)
    from memo_amd import synth
    n_docs, length, k = 500, 1_000_000, 31
    ix, (r0, r1) = synth.device_index(0, length, k, n_docs, length, pack="only")
    with ix:
        v = ix.conservation(0, length, k, n_docs)
    assert v.dtype == np.uint16 and v.max() > 255
    b_want, blocks = wanted(v)
    got, taken, room, size = round_trip(v, b_want)
    assert np.array_equal(got, v) and taken == b_want
    assert size < 0.45 * length, size                                             # < 3.6 bits per position on the wire (16 plain)
    with pytest.raises(memo.MemoError):
        _lib.check(L.memo_transport_runs16_pack_dev(1, 100, 6, 16, 0, None))     # capacity not a multiple of 4


def test_transport_dense_coding_round_trip(memo):
    """uint8 results -> 2 bits + escape nibbles (allocated exactly per 32768 positions) + exception list
    -> uint8: geometric value mixes (what conservation looks like), all-escape and no-escape data,
    ragged lengths around the 4096-position rounds and 32768-position blocks, capacities that do not
    suffice (reported, never silent)"""
    import ctypes as C
    import torch
    from memo_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(16)

    def round_trip(v, b_cap, cap):
        n = len(v)
        src = torch.from_numpy(v).cuda() if n else torch.zeros(16, dtype=torch.uint8, device="cuda")
        wire = torch.zeros(L.memo_transport_dense_bytes(n, b_cap, cap), dtype=torch.uint8, device="cuda")
        dst = torch.full((max(n, 16),), 77, dtype=torch.uint8, device="cuda")
        _lib.check(L.memo_transport_dense_pack_dev(src.data_ptr(), n, b_cap, cap, wire.data_ptr(), 0, None))
        found, have, taken, room = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(L.memo_transport_dense_stats(wire.data_ptr(), 0, None, C.byref(found), C.byref(have),
                                                C.byref(taken), C.byref(room)))
        _lib.check(L.memo_transport_dense_unpack_dev(wire.data_ptr(), n, b_cap, cap, dst.data_ptr(), 0, None))
        torch.cuda.synchronize()
        return dst[:n].cpu().numpy(), found.value, have.value, taken.value, room.value, wire.numel()

    for n in (0, 1, 15, 16, 17, 4095, 4096, 4097, 12345, 32767, 32768, 32769, 1_000_003, 3_000_000):
        for mix in ("geometric", "uniform", "ones", "zeros", "big"):
            if mix == "geometric":
                v = np.minimum(rng.geometric(0.33, n), 200).astype(np.uint8)
            elif mix == "uniform":
                v = rng.integers(0, 256, n).astype(np.uint8)
            elif mix == "ones":
                v = rng.integers(1, 4, n).astype(np.uint8)
            elif mix == "zeros":
                v = np.zeros(n, np.uint8)
            else:
                v = rng.integers(18, 256, n).astype(np.uint8)
            escapes = ~((v >= 1) & (v <= 3))
            per_block = [int(escapes[i:i + 32768].sum()) for i in range(0, n, 32768)]
            b_want = sum((((c + 1) // 2) + 3) & ~3 for c in per_block)
            exc_want = int((v > 17).sum())
            got, found, have, taken, room, size = round_trip(v, b_want, max(exc_want, 4))
            assert (found, have, taken, room) == ((exc_want, max(exc_want, 4), b_want, b_want) if n else (0, 0, 0, 0)), (n, mix)
            assert np.array_equal(got, v), (n, mix)
            blocks, chunks = len(per_block), (n + 4095) // 4096
            assert size == ((((16 + 8 * blocks + 1024 * chunks + b_want + 7) & ~7) + 8 * max(exc_want, 4) + 15) & ~15)
            if mix == "geometric" and n >= 1_000_000:
                assert size < 0.46 * n                             # 3.7 bits per position at this mix
    # capacities that do not suffice are reported, not silently accepted
    v = rng.integers(18, 256, 100_000).astype(np.uint8)
    got, found, have, taken, room, size = round_trip(v, 50_016, 10)
    assert found == 100_000 and have == 10 and taken <= room
    got, found, have, taken, room, size = round_trip(v, 20_000, 100_000)
    assert taken > room and found == 100_000
    with pytest.raises(memo.MemoError):
        round_trip(v, 6, 10)                                       # not a multiple of 4
