"""Index-row construction: the oracle restatement against the reference's dap_to_bed.py output."""
import json
import os

import pytest

from tests import golden_util as G

DAP = os.path.join(G.GOLD, "dap")
CASES = json.load(open(os.path.join(DAP, "manifest.json")))


@pytest.mark.parametrize("c", CASES, ids=lambda c: c["name"])
def test_dap_oracle_matches_reference(c):
    from oracle import dap_oracle as D
    names, rec_begin = D.read_fai(os.path.join(DAP, c["fai"]))
    _, lcp = D.read_dap(os.path.join(DAP, c["dap"]))
    rows = D.dap_rows(lcp, rec_begin, c["overlap"], c["order"])
    text = D.bed_text(names, *rows)
    assert G.sha(text.encode()) == c["sha256"]
    assert text == open(os.path.join(DAP, c["name"] + ".bed")).read()
