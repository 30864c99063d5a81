"""Every emission form and the residual skip index, forced through the C ABI and checked against the CPU oracle.

The default policy picks the row-kernel variant per graph and the emission form per row, so the ordinary parity tests do not
reach every combination on their small graphs; here the debug switches (BVG_EMIT, BVG_DBG 16/32, BVG_NOSKIP — read at
every call, live because tests/conftest.py sets BVG_TEST_KNOBS) pin each one: level-synchronous tasks on every row, the pipelined
loop inside the task variant, the pipelined variant alone, all of them with and without the skip index, and the lean scan kernel
(csrc/bvg_scan.hip) that takes the validated blocks of an indexed scan."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = {
    # the row kernel's own scan paths (BVG_SCANK=0 keeps the lean scan kernel away from the validated blocks)
    "tasks_every_row": dict(BVG_EMIT="1", BVG_DBG="16", BVG_SCANK="0"),
    "per_row_choice": dict(BVG_EMIT="1", BVG_SCANK="0"),
    "pipelined_rows_in_task_variant": dict(BVG_EMIT="1", BVG_DBG="32", BVG_SCANK="0"),
    "pipelined_variant": dict(BVG_EMIT="0"),
    "tasks_no_skip_index": dict(BVG_EMIT="1", BVG_DBG="16", BVG_NOSKIP="1"),
    "pipelined_no_skip_index": dict(BVG_EMIT="0", BVG_NOSKIP="1"),
    # the lean scan kernel (csrc/bvg_scan.hip) on every validated block: its default pool, a pool so small that rows are cut all the
    # time and blocks fail over, and the validating pass forced onto the pipelined rows (which validate nothing: no block is lean)
    "scan_kernel": dict(BVG_EMIT="1"),
    "scan_kernel_small_pool": dict(BVG_EMIT="1", BVG_SCAN_POOL="704"),
    "scan_kernel_after_pipelined_validation": dict(BVG_EMIT="1", BVG_DBG="32"),
}
KNOBS = ("BVG_EMIT", "BVG_DBG", "BVG_NOSKIP", "BVG_SCANK", "BVG_SCAN_POOL")


@pytest.fixture(params=sorted(MODES))
def mode(request, monkeypatch):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in MODES[request.param].items():
        monkeypatch.setenv(k, v)
    return request.param


def _oracle_graph(O, st):
    return O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)


def _check(W, O, st, ranges=()):
    n = st.params.nodes
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = _oracle_graph(O, st)
    r, o = g.scan(), og.scan()                                        # also builds the skip index when it is enabled
    assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])
    deg, succ = g.decode_range(0, n)
    odeg, osucc = og.decode_range(0, n)
    assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc)
    for a, b in ranges:
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), (a, b)
    r2 = g.scan()                                                     # steady state: index present, tiers learned
    assert (r2["nodes"], r2["arcs"], r2["chk"]) == (o["nodes"], o["arcs"], o["chk"])
    r["lean_blocks_steady"] = r2["lean_blocks"]
    g.close()
    return r


def test_dense_copy_heavy_graph(W, tools, oracle, mode):
    """eu-2015-shaped: long lists, long residual lists at the roots of the reference chains (skip segments), 4 levels per row."""
    st = tools.synth_store(20000, seed=11, synth=tools.eu_like(), threads=4)
    r = _check(W, oracle, st, ranges=[(0, 1), (5000, 20000), (4097, 4099), (19999, 20000), (6000, 6000)])
    if "no_skip" not in mode:
        assert r["index_bytes"] > 8 * 20001 + 20 * 400, "the skip index should have been built and counted"
    if mode in ("scan_kernel", "scan_kernel_small_pool"):
        assert r["lean_blocks_steady"] > 50, "the lean scan kernel should have taken the validated blocks"
    else:
        assert r["lean_blocks_steady"] == 0, mode


def test_sparse_graph_with_reference_chains(W, tools, oracle, mode):
    st = tools.synth_store(30000, seed=12, synth=tools.web_like(), threads=4)
    _check(W, oracle, st, ranges=[(123, 29000)])


def test_reference_free_graph(W, tools, oracle, mode):
    st = tools.synth_store(30000, seed=13, params=W.default_params(window_size=0, max_ref_count=0, min_interval_length=0), synth=tools.web_like(), threads=4)
    _check(W, oracle, st)


def test_heavy_tail_and_giant_lists(W, tools, oracle, mode):
    """Lists of thousands of successors: big-LDS classes and the global-memory tier next to the task rows."""
    st = tools.synth_store(12000, seed=14, synth=tools.eu_like(max_deg=30000, tail_alpha=1.6, mean_deg=40.0), threads=4)
    _check(W, oracle, st)


@pytest.mark.parametrize("params", [
    dict(outdegree_coding=1, block_coding=1, residual_coding=1, reference_coding=1, block_count_coding=1),
    dict(residual_coding=7, block_count_coding=5, block_coding=5), dict(residual_coding=3, zeta_k=5), dict(zeta_k=1),
    dict(window_size=16, max_ref_count=1000, min_interval_length=2),
])
def test_other_codings_and_parameters(W, tools, oracle, mode, params):
    st = tools.synth_store(8000, seed=15, params=W.default_params(**params), synth=tools.eu_like(mean_deg=50.0), threads=4)
    _check(W, oracle, st)


def test_wide_ids_and_node_base(W, tools, oracle, mode):
    st = tools.synth_store(6000, seed=16, synth=tools.eu_like(), threads=2)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = _oracle_graph(oracle, st)
    for base, wide in ((0xFFFFF000, False), ((1 << 33) + 7, False), ((1 << 40) + 1, True)):
        g.set_node_base(base)
        g.set_tuning(force_wide=wide)
        assert g.scan()["chk"] == og.scan(0, 6000, node_base=base)["chk"], (base, wide)
    deg, succ = g.decode_range(0, 6000)
    odeg, osucc = og.decode_range(0, 6000)
    assert np.array_equal(succ, osucc + ((1 << 40) + 1))
    g.close()


def test_randomised_shapes(W, tools, oracle, mode):
    rng = np.random.default_rng(77)
    for trial in range(12):
        kw = dict(window_size=int(rng.choice([0, 1, 7, 16])), max_ref_count=int(rng.choice([0, 1, 3, 1000])),
                  min_interval_length=int(rng.choice([0, 2, 4])), zeta_k=int(rng.choice([2, 3, 5])))
        n = int(rng.choice([65, 700, 5000, 9000]))
        synth = tools.web_like(mean_deg=float(rng.choice([3, 30, 120])), p_copy=float(rng.choice([0.0, 0.6, 0.95])), p_empty=float(rng.choice([0.0, 0.3])),
                               p_interval=float(rng.choice([0.0, 0.6])), max_deg=int(rng.choice([50, 2000, 9000])), window=int(rng.choice([1, 7])))
        st = tools.synth_store(n, seed=int(rng.integers(1 << 30)), params=W.default_params(**kw), synth=synth, chunk_nodes=1 << 12, threads=2)
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
        og = _oracle_graph(oracle, st)
        r, o = g.scan(), og.scan()
        assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"]), (trial, kw, n)
        deg, succ = g.decode_range(0, n)
        odeg, osucc = og.decode_range(0, n)
        assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc), (trial, kw, n)
        g.close()


def test_long_records(W, tools, oracle, monkeypatch):
    """Records of 32-160 Kbit (thousands of far residuals) exceed the 4 KiB LDS stream window: the plan files their blocks under the
    global-memory kernel; lists copying every other element of such a list overflow the copy-block scratch of the LDS classes and
    cascade there too.  Whichever tier ends up decoding them, the result must agree with the oracle."""
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(5)
    n = 1 << 23                                                              # a wide id space makes the gaps (and the codes) long
    rows = {}
    for x in range(7, n, 600000):
        big = np.unique(rng.integers(0, n, size=4500))                        # ~4 500 residuals x ~15 bits: ~65 Kbit
        rows[x] = big
        rows[x + 1] = np.unique(np.concatenate([big[::2], rng.integers(0, n, size=5)]))   # copies half of it
    rows[1000001] = np.unique(rng.integers(0, n, size=10500))                 # ~10 500 residuals (still an LDS-sized list): > 128 Kbit
    for x in rng.integers(0, n, size=20000):
        rows.setdefault(int(x), np.unique(rng.integers(max(0, x - 50), min(n, x + 50), size=int(rng.integers(1, 8)))))
    deg0 = np.zeros(n, dtype=np.int64)
    for x, l in rows.items():
        deg0[x] = len(l)
    off = np.concatenate([[0], np.cumsum(deg0)]).astype(np.uint64)
    adj = np.concatenate([rows[x] for x in sorted(rows)]).astype(np.int64)
    st = tools.store((off, adj), W.default_params(), threads=4)
    rec_bits = np.diff(st.offsets.astype(np.int64))
    assert (rec_bits > 32768).sum() >= 10 and rec_bits.max() > 131072
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = _oracle_graph(oracle, st)
    for _ in range(2):                                                       # second scan: learned tiers
        r, o = g.scan(), og.scan()
        assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])
    deg, succ = g.decode_range(0, n)
    assert np.array_equal(deg, deg0) and np.array_equal(succ, adj)
    g.close()
