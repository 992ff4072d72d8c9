import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
np.seterr(over="ignore")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle and (if needed) the product library once per session."""
    import oracle
    oracle.build()
    lib = os.path.join(ROOT, "swipe_amd", "libswipe_amd.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()


def load_golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
        return json.load(f)


def case_matrix(case, mod):
    """matrix (int64[1024]) of a case through module `mod` (oracle or swipe_amd)."""
    if case.sym == 0:
        return mod.matrix_nucleotide(case.match, case.mismatch)
    if case.matrix == "@text":
        return mod.matrix_parse(case.matrix_text)
    return mod.matrix_builtin(case.matrix)
