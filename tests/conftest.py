import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# `memo query` leaves a sidecar cache next to the index it read and builds it in a detached process; the tests
# that are about the cache switch it on themselves
os.environ.setdefault("MEMO_CACHE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import memo_oracle
    memo_oracle.lib()
    return memo_oracle


@pytest.fixture(scope="session")
def oracle_mod(oracle):
    return oracle
