"""The flat scan kernel (webgraph-big_amd/experimental/bvg_flat.hip, round 5) on the GPU: built only by `make experimental`, selected by BVG_FLAT=1; scans of a dense, a
sparse and the reference's own graph against the oracle, in a child process (the library is chosen when the package first loads it).  Slower than scan_kernel
(DESIGN.md), kept bit-exact."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "webgraph-big_amd", "lib", "libbvgraph_hip_experimental.so")


@pytest.mark.gpu
@pytest.mark.parametrize("recs,shape,n,seed", [(64, "eu", 60000, 5), (128, "web", 200000, 3), (256, "cnr", 325557, 0)])
def test_flat_scan_kernel_matches_the_oracle(recs, shape, n, seed):
    if not os.path.exists(LIB):
        pytest.skip("experimental library not built (make -C webgraph-big_amd experimental)")
    e = dict(os.environ, BVG_EMU_LIB=LIB, BVG_TEST_KNOBS="1", BVG_FLAT="1", BVG_FLAT_RECS=str(recs), BVG_DEBUG="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_case.py"), str(n), str(seed), shape, "3"], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "emu case ok" in out.stdout and "flat kernel:" in out.stderr          # the flat kernel really ran (bvg_api.hip prints its geometry)
    assert "lean_blocks 0 " not in out.stdout.splitlines()[2]


@pytest.mark.gpu
@pytest.mark.parametrize("dbg,shape,n,seed", [(4096, "eu", 60000, 5), (8192, "eu", 60000, 5), (8192, "cnr", 325557, 0), (12288, "web", 200000, 3)])
def test_round6_list_builds_match_the_oracle(dbg, shape, n, seed):
    """Round 6's two structural experiments inside scan_kernel, compiled into the experimental build only (they lost: csrc/bvg_scan.hip, "MEASURED"): WW (BVG_DBG=4096: a stored
    list with reference built wave-wide from lane bit vectors -- copy mask by prefix-XOR, compaction, in-place spread) and ZE (8192: position tasks by kept element over
    the extras' bit vectors), alone and together, against the oracle on a dense, a sparse and the reference's own graph."""
    if not os.path.exists(LIB):
        pytest.skip("experimental library not built (make -C webgraph-big_amd experimental)")
    e = dict(os.environ, BVG_EMU_LIB=LIB, BVG_TEST_KNOBS="1", BVG_DBG=str(dbg))
    e.pop("BVG_FLAT", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_case.py"), str(n), str(seed), shape, "3"], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "emu case ok" in out.stdout and "lean_blocks 0 " not in out.stdout.splitlines()[2]
