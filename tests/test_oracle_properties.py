"""Property tests of the oracle's three restatements against each other (hypothesis).  CPU-only."""
import numpy as np
from hypothesis import given, settings, strategies as st


@st.composite
def queries(draw):
    n_docs = draw(st.integers(2, 70))
    m = draw(st.integers(0, 60))
    length = draw(st.integers(1, 200))
    s = np.sort(np.array(draw(st.lists(st.integers(-5, length + 40), min_size=m, max_size=m)), np.int64))
    ln = np.array(draw(st.lists(st.integers(0, 140), min_size=m, max_size=m)), np.int64)
    o = np.array(draw(st.lists(st.integers(1, n_docs - 1), min_size=m, max_size=m)), np.int64)
    qs = draw(st.integers(0, length))
    qe = draw(st.integers(qs, length + 30))
    k = draw(st.integers(1, 130))
    return s, s + ln, o, qs, qe, k, n_docs


@settings(max_examples=300, deadline=None)
@given(queries())
def test_literal_closed_numpy_agree(oracle_mod, q):
    s, e, o, qs, qe, k, n = q
    a = oracle_mod.conservation(s, e, o, qs, qe, k, n, literal=True)
    assert np.array_equal(a, oracle_mod.conservation(s, e, o, qs, qe, k, n, literal=False))
    assert np.array_equal(a, oracle_mod.np_conservation(s, e, o, qs, qe, k, n))
    # filter_pq drops only rows that cannot write
    assert np.array_equal(a, oracle_mod.conservation(*oracle_mod.filter_rows(s, e, o, qs, qe, k), qs, qe, k, n))
    b = oracle_mod.membership(s, e, o, qs, qe, k, n, literal=True)
    assert np.array_equal(b, oracle_mod.membership(s, e, o, qs, qe, k, n, literal=False))
    assert np.array_equal(b, oracle_mod.np_membership(s, e, o, qs, qe, k, n))
    # pivot column is never written (annot >= 1); conservation >= 1; bits beyond N are zero
    if qe > qs:
        assert np.all(b[:, 0] & 1) and a.min() >= 1 and a.max() <= n
        assert np.all((b[:, -1] >> np.uint32((n - 1) % 32)) <= 1)
    # min over covering rows == first genome column cleared, on a monotone index only; in general
    # conservation value is the smallest annot whose membership bit is cleared
    mat = oracle_mod.bits_to_matrix(b, n)
    first_clear = np.where((mat == 0).any(1), (mat == 0).argmax(1), n)
    assert np.array_equal(first_clear.astype(np.uint16), a)


@settings(max_examples=100, deadline=None)
@given(queries(), st.integers(1, 6))
def test_window_split_property(oracle_mod, q, parts):
    s, e, o, qs, qe, k, n = q
    whole = oracle_mod.conservation(s, e, o, qs, qe, k, n, literal=False)
    cuts = np.linspace(qs, qe, parts + 1).astype(int)
    got = [oracle_mod.conservation(*oracle_mod.filter_rows(s, e, o, int(a), int(b), k), int(a), int(b), k, n, literal=False)
           for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(got) if got else whole, whole)
