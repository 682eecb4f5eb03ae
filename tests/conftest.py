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
    # Load order: the torch wheel bundles its own ROCm runtime, libmemo_amd.so links the system's.  Whichever HIP
    # runtime a process loads first serves both; with the system's first, torch then finds no GPU ("No HIP GPUs are
    # available") -- seen when one GPU test module ran alone and its first torch user came after the library.  Same
    # order as bench.py and the tools: torch first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import memo_oracle
    memo_oracle.lib()
    return memo_oracle


@pytest.fixture(scope="session")
def oracle_mod(oracle):
    return oracle
