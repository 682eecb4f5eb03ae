"""Pins the CPU oracle (oracle/) against the golden vectors produced by running the
reference (tools/make_golden.py).  CPU-only."""
import numpy as np
import pytest

from tests import golden_util as G

OK = G.cases(raises=False)
BAD = G.cases(raises=True)


def test_manifest_shape():
    assert len(OK) > 250 and len(BAD) == 2
    assert any(c["membership"] for c in OK) and any(not c["membership"] for c in OK)


@pytest.mark.parametrize("c", OK, ids=lambda c: c["name"])
def test_oracle_matches_reference(c, oracle):
    rec, qs, qe = G.region(c)
    z = G.load(c)
    s, e, o = G.index_columns(c["index"], rec)
    # filter_pq restatement returns exactly the rows (and row order) the reference got
    fs, fe, fo = oracle.filter_rows(s, e, o, qs, qe, c["k"])
    ref_rows = z["rows"].astype(np.int64).reshape(-1, 3)
    assert np.array_equal(np.stack([fs, fe, fo], 1).reshape(-1, 3), ref_rows)
    for rows in ((fs, fe, fo), (s, e, o)):          # filtered rows, and the whole chromosome
        if rows[0] is s and not np.all(e >= s):
            continue
        if c["membership"]:
            want = G.expected_matrix(c, z)
            for got in (oracle.membership(*rows, qs, qe, c["k"], c["n"], literal=True),
                        oracle.membership(*rows, qs, qe, c["k"], c["n"], literal=False),
                        oracle.np_membership(*rows, qs, qe, c["k"], c["n"])):
                assert np.array_equal(oracle.bits_to_matrix(got, c["n"]), want)
                text = oracle.emit_membership(got, c["n"])
        else:
            want = z["vec"]
            for got in (oracle.conservation(*rows, qs, qe, c["k"], c["n"], literal=True),
                        oracle.conservation(*rows, qs, qe, c["k"], c["n"], literal=False),
                        oracle.np_conservation(*rows, qs, qe, c["k"], c["n"])):
                assert np.array_equal(got.astype(np.int64), want)
                text = oracle.emit_conservation(got)
        assert G.sha(text) == c["sha256"]
        if "out" in c:
            assert text == G.out_bytes(c)


@pytest.mark.parametrize("c", BAD, ids=lambda c: c["name"])
def test_oracle_index_error(c, oracle):
    rec, qs, qe = G.region(c)
    s, e, o = G.index_columns(c["index"], rec)
    assert c["raises"] == "IndexError"
    fn = oracle.membership if c["membership"] else oracle.conservation
    npf = oracle.np_membership if c["membership"] else oracle.np_conservation
    for lit in (True, False):
        with pytest.raises(IndexError):
            fn(s, e, o, qs, qe, c["k"], c["n"], literal=lit)
    with pytest.raises(IndexError):
        npf(s, e, o, qs, qe, c["k"], c["n"])


def test_window_split_equals_whole(oracle):
    """Sharding property: per-sub-window runs concatenate to the full run."""
    s, e, o = G.index_columns("rnd_n40.parquet", "chr1")
    qs, qe, k, n = 100, 4100, 31, 40
    whole = oracle.conservation(s, e, o, qs, qe, k, n, literal=False)
    cuts = [qs, 777, 1500, 1501, 3000, qe]
    parts = [oracle.conservation(*oracle.filter_rows(s, e, o, a, b, k), a, b, k, n, literal=False)
             for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts), whole)


def test_synth_rows_addressable(oracle):
    s, e, o = oracle.synth_rows(0, 5000, 5, 1, 100)
    s2, e2, o2 = oracle.synth_rows(1234, 100, 5, 1, 100)
    assert np.array_equal(s[1234:1334], s2) and np.array_equal(e[1234:1334], e2) and np.array_equal(o[1234:1334], o2)
    assert np.all(np.diff(s) >= 0) and s[0] == 1 and np.all(e >= s) and np.all((e - s) < 60)
    assert o.min() >= 1 and o.max() <= 99
    sh, _, _ = oracle.synth_rows(0, 100, 1, 2, 10)      # 0.5 rows per position
    assert np.array_equal(sh, 1 + 2 * np.arange(100))


def _view_cases():
    import json
    import os
    return json.load(open(os.path.join(G.GOLD, "view", "manifest.json")))


@pytest.mark.parametrize("c", _view_cases(), ids=lambda c: c["name"])
def test_view_table_matches_reference(c, oracle):
    import os
    z = np.load(os.path.join(G.GOLD, "view", c["name"] + ".npz"))
    if "raises" in c:
        with pytest.raises(ZeroDivisionError):
            oracle.view_table(z["vec"], c["n_docs"], c["n_bins"])
        return
    got = oracle.view_table(z["vec"], c["n_docs"], c["n_bins"])
    assert np.array_equal(got["bin"], z["bin"]) and np.array_equal(got["No. Genomes"], z["genomes"])
    assert np.array_equal(got["value"], z["value"])          # float64, bit for bit
