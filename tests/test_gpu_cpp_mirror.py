"""GPU: the C++ host mirror (webgraph-big_amd/host/bvgraph.hpp) driven by a compiled C++ program."""
import os
import re
import subprocess

import pytest

from conftest import ROOT, CNR

pytestmark = pytest.mark.gpu


def test_cpp_mirror_scans_cnr2000(W, oracle):
    exe = os.path.join(ROOT, "webgraph-big_amd", "lib", "test_host_mirror")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "webgraph-big_amd"), "lib/test_host_mirror"])
    out = subprocess.run([exe, CNR], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"OK nodes=(\d+) arcs=(\d+) chk=([0-9a-f]+) scan_arcs=(\d+) scan_chk=([0-9a-f]+) split_arcs=(\d+)", out.stdout)
    assert m, out.stdout
    o = oracle.Graph.load(CNR).scan()
    assert int(m.group(1)) == 325557
    assert "STORE 1198480 bytes identical" in out.stdout or re.search(r"STORE \d+ bytes identical", out.stdout), out.stdout   # bvg_store through the C++ mirror regenerates cnr-2000.graph
    assert int(m.group(2)) == int(m.group(4)) == int(m.group(6)) == o["arcs"] == 3216152
    assert int(m.group(3), 16) == int(m.group(5), 16) == o["chk"]


def test_cpp_mirror_reads_labelled_graph(W, tools, tmp_path):
    """BitStreamArcLabelledImmutableGraph of the C++ mirror on files written by the tooling (labels = f(source, position))."""
    import numpy as np
    exe = os.path.join(ROOT, "webgraph-big_amd", "lib", "test_host_mirror")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "webgraph-big_amd"), "lib/test_host_mirror"])
    n = 4000
    st = tools.synth_store(n, seed=3, threads=2)
    off, adj = tools.synth_adjacency(n, seed=3)
    deg = np.diff(off.astype(np.int64))
    src = np.repeat(np.arange(n, dtype=np.int64), deg)
    pos = np.arange(int(off[-1]), dtype=np.int64) - np.repeat(off[:-1].astype(np.int64), deg)
    vals = ((src * 31 + pos) & 1023).astype(np.int32)
    st.write(str(tmp_path / "under"))
    tools.store_labels(2, 10, vals, off).write(str(tmp_path / "lab"), "under")
    out = subprocess.run([exe, str(tmp_path / "under"), str(tmp_path / "lab")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "LABELS %d ok" % len(vals) in out.stdout, out.stdout + out.stderr
