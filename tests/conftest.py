import gzip
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the mode-forcing switches of the product library (BVG_EMIT, BVG_DBG, BVG_GIANT, BVG_NOSKIP, ...) are live only in a process
# that had BVG_TEST_KNOBS set when the library was first used (csrc/bvg_kernels.h: knob())
os.environ.setdefault("BVG_TEST_KNOBS", "1")

GOLDEN = os.path.join(ROOT, "tests", "golden")
CNR = os.path.join(GOLDEN, "cnr-2000")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a GPU test that stops (a kernel that never drains, a host thread that never returns) should end with the stacks of every
    # thread on stderr instead of holding the box until the caller's limit kills the run without a trace
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for it in items:
        if "gpu" in it.keywords and it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(600, method="thread"))


@pytest.fixture(scope="session")
def cnr_golden():
    """The reference's own expected answer (BVGraphTest.testLarge): list of int64 arrays, one per node."""
    with gzip.open(CNR + ".graph-txt.gz", "rb") as f:
        lines = f.read().split(b"\n")
    n = int(lines[0])
    lists = [np.array(l.split(), dtype=np.int64) for l in lines[1:n + 1]]
    return lists


@pytest.fixture(scope="session")
def cnr_csr(cnr_golden):
    deg = np.array([len(a) for a in cnr_golden], dtype=np.int32)
    succ = np.concatenate(cnr_golden)
    return deg, succ


@pytest.fixture(scope="session")
def oracle():
    from oracle import bvg_oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def tools():
    import tooling as T
    T.lib()
    return T


@pytest.fixture(scope="session")
def W():
    import webgraph_big_amd as W
    return W
