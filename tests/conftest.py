import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hg():
    """The product package; building the HIP library if it is missing."""
    import hypergef_amd
    from hypergef_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return hypergef_amd


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def vertex_csr(inc, oracle):
    """H CSR from H_T CSR through the oracle's transpose."""
    return oracle.transpose_csr(inc.M, inc.N, inc.csrptr, inc.colind)
