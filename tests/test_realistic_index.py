"""An index built from SEQUENCES (tools/realistic_index.py: a random pivot, mutated copies with SNPs / indels /
rearrangements, matching statistics by suffix automaton, dap_to_bed on the GPU) instead of the uniform generator:
whole-window parity of the HIP sweeps with the oracle on every row format, at k = 21 / 31 / 101, both queries.
Round-2 VERDICT item 6: rows per position, overlap lengths and annots here are what a real `memo index` produces
(orders are monotone, overlaps are long, positions without rows are common), not the generator's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def real_index(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("real"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "realistic_index.py"), "--length", "600000", "--genomes", "16",
                        "--out", out, "--threads", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    stats = json.loads(r.stdout.strip().splitlines()[-1])
    return out, stats


@pytest.mark.parametrize("query", ["cons", "memb"])
def test_sequence_built_index_whole_window(real_index, query, oracle):
    import memo_amd
    out, stats = real_index
    z = np.load(os.path.join(out, query + ".npz"))
    s, e, o, n, L = z["start"], z["end"], z["annot"], int(z["num_docs"]), int(z["length"])
    assert stats[query]["start_sorted"] and len(s) == stats[query]["rows"] and (e >= s).all()
    if query == "cons":                                            # orders of a real index are monotone per position:
        assert o.min() >= 1 and o.max() <= n - 1                   # (dap_to_bed.py:89-90: rank after a descending sort)
    fn = oracle.membership if query == "memb" else oracle.conservation
    families = {}
    for k in (21, 31, 101):
        want = fn(*oracle.filter_rows(s, e, o, 0, L, k), 0, L, k, n, literal=False)
        if query == "cons":                                        # what the values MEAN on a real index: never more genomes
            assert want.min() >= 1 and want.max() <= n             # than there are, the pivot always counted
        ways = [("int64", lambda: memo_amd.DeviceIndex.from_host(s, e, o)),
                ("packed", lambda: memo_amd.DeviceIndex.from_host_packed(s, e, o))]
        from memo_amd.index import dense_rows_can_answer
        if dense_rows_can_answer(len(s), int(s[0]), int(s[-1]), int(o.max()), k, n, query == "memb"):
            ways.append(("dense", lambda: memo_amd.DeviceIndex.from_host_packed(s, e, o, dense=True)))
        for name, make in ways:
            with make() as ix:
                got = ix.membership(0, L, k, n) if query == "memb" else ix.conservation(0, L, k, n)
                assert np.array_equal(got, want), (query, k, name)
                families[(k, name)] = ix.info()["last_sweep"]
                a, b = L // 3 + 5, L // 3 + 70_001                 # a sub-window is the slice of the whole window
                sub = ix.membership(a, b, k, n) if query == "memb" else ix.conservation(a, b, k, n)
                assert np.array_equal(sub, want[a:b]), (query, k, name)
        one = (memo_amd.membership if query == "memb" else memo_amd.conservation)(s, e, o, 1000, L - 1000, k, n)
        assert np.array_equal(one, want[1000:L - 1000]), (query, k, "one-shot")
    print(query, "rows/position %.2f" % stats[query]["rows_per_position"], "kernel families:", families)
