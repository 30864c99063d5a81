"""CPU: the oracle (oracle/bvg_oracle.c) and the tooling (tooling/bvg_store.cpp: the encoder every synthetic input comes from) built with
AddressSanitizer + UndefinedBehaviorSanitizer and driven through their whole surface in a child process: encode synthetic graphs with
every coding, re-store the reference's fixture, decode sequentially / by random access / from arbitrary starts, scan with threads,
labels, hand-assembled odd records.  Any report fails the test (sanitizer-clean CPU code; GPU sanitizers do not exist on
this pool)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SCRIPT = r"""
import gzip, os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import webgraph_big_amd as W
import tooling as T
from oracle import bvg_oracle as O
CNR = os.path.join(%(root)r, "tests", "golden", "cnr-2000")
# the reference fixture: decode, re-store byte for byte
og = O.Graph.load(CNR)
n = og.num_nodes()
deg, succ = og.decode_range(0, n)
assert int(deg.sum()) == 3216152
cum = np.concatenate([[0], np.cumsum(deg, dtype=np.int64)]).astype(np.uint64)
st = T.store((cum, succ), W.default_params(min_interval_length=3), threads=2)
assert st.graph.tobytes() == open(CNR + ".graph", "rb").read()
for x in (0, 1, 777, n - 1):
    assert og.successors(x).tolist() == succ[int(cum[x]):int(cum[x + 1])].tolist()
assert og.scan(threads=3)["arcs"] == 3216152
# every coding / parameter family through the encoder and back
rng = np.random.default_rng(1)
for kw in (dict(), dict(window_size=0, max_ref_count=0, min_interval_length=0), dict(residual_coding=1, outdegree_coding=1, reference_coding=2, block_count_coding=5, block_coding=1),
           dict(residual_coding=7), dict(residual_coding=3, zeta_k=5), dict(zeta_k=1, window_size=20, max_ref_count=-1, min_interval_length=2), dict(window_size=70, max_ref_count=5)):
    s2 = T.synth_store(3000, seed=int(rng.integers(1 << 20)), params=W.default_params(**kw), synth=T.eu_like(mean_deg=30.0), chunk_nodes=512, threads=2)
    o2 = O.Graph.from_memory(O.Params(**s2.params.as_dict()), s2.graph.tobytes(), s2.offsets)
    d2, a2 = o2.decode_range(0, 3000)
    off, adj = T.synth_adjacency(3000, seed=0, synth=T.web_like(), chunk_nodes=512)
    assert int(d2.sum()) == s2.stats["arcs"]
    o2.scan(100, 2900, threads=2); o2.decode_range(1234, 1300)
    T.encode_offsets(s2.offsets, 2); T.encode_offsets(s2.offsets, 1)
# labels
vals = rng.integers(0, 1000, size=int(d2.sum())).astype(np.int32)
aoff = np.concatenate([[0], np.cumsum(d2, dtype=np.int64)]).astype(np.uint64)
T.store_labels(1, 0, vals, aoff); T.store_labels(2, 10, vals, aoff)
# hand-assembled odd records (streams the encoder never writes: tests/test_malformed_streams.py)
import test_malformed_streams as M
for case in sorted(M.CASES):
    p, g, offs, lists = M._store(W, M._graph(case, 3, 5))
    o3 = O.Graph.from_memory(O.Params(**p.as_dict()), g.tobytes(), offs)
    o3.decode_range(0, p.nodes); [o3.successors(x) for x in range(p.nodes)]; o3.scan()
print("SANITIZED-OK")
"""


def _asan_lib(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_and_tooling_are_sanitizer_clean():
    asan = _asan_lib("libasan.so")
    if not asan:
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tooling"), "asan"])
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=87",
               BVG_ORACLE_LIB=os.path.join(ROOT, "oracle", "libbvg_oracle_asan.so"), BVG_TOOLS_LIB=os.path.join(ROOT, "tooling", "lib", "libbvg_tools_asan.so"))
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
