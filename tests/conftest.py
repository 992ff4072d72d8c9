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
    config.addinivalue_line("markers", "late: written after the round's last run on hardware; collected after every other test, so "
                                       "that under -x a surprise in one of these does not hide the tests that are known to pass")


def pytest_collection_modifyitems(config, items):
    items.sort(key=lambda it: it.get_closest_marker("late") is not None)    # stable: order within each class is kept


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


def shard_devices(k):
    """HIP devices for k shards.  SWA_TEST_DEVICES (comma-separated ordinals) names the devices to use; default: every
    device the library sees.  The list is cycled, so a box with ONE MI355X runs all shards on device 0 (k handles, k host
    threads, k sets of streams) and a box with more devices - or one partitioned into several - exercises distinct ordinals
    with no change to the tests."""
    import swipe_amd
    env = os.environ.get("SWA_TEST_DEVICES", "").strip()
    pool = [int(x) for x in env.split(",") if x.strip() != ""] if env else list(range(max(1, swipe_amd._lib.load().swa_device_count())))
    return [pool[i % len(pool)] for i in range(k)]


def under_interpreter():
    """the suite is running on tools/gfx950sim (LD_PRELOAD, no GPU): kernels are interpreted, every device access is checked"""
    return os.environ.get("HIPSIM") == "1"


def interpreter_clear_fault():
    """a device fault is sticky, as on hardware; a test that provokes one on purpose clears it for the tests after it"""
    if under_interpreter():
        import ctypes
        ctypes.CDLL(None).hipsim_clear_fault()
