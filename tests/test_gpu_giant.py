"""The giant-list kernel (csrc/bvg_giant.hip, tier 2a): lists and records too large for the LDS row kernels, decoded by a whole
workgroup each (header / copy blocks / intervals in step from a sliding window, residuals cut at skip-index entries, emission by
output position).  Checked against the CPU oracle through the C ABI: forced onto EVERY block of ordinary graphs (BVG_GIANT=2), on
graphs that really hold giant lists (with and without the residual skip index, scan and materialise), and on the hand-assembled
streams whose parts overlap, which it must hand over to the generic kernel (MergedLongIterator.java:85-89)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_graph(O, st):
    return O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)


def _check(W, O, st, ranges=(), scans=2):
    n = st.params.nodes
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = _oracle_graph(O, st)
    o = og.scan()
    res = []
    for _ in range(scans):                                             # the first scan builds the skip index, the second uses it
        r = g.scan()
        assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])
        res.append(r)
    deg, succ = g.decode_range(0, n)
    odeg, osucc = og.decode_range(0, n)
    assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc)
    for a, b in ranges:
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), (a, b)
    g.close()
    return res


@pytest.fixture
def every_block(monkeypatch):
    monkeypatch.setenv("BVG_GIANT", "2")
    monkeypatch.setenv("BVG_DEBUG", "1")                      # names the tiers on stderr: see _none_left_to_the_generic_kernel


def _none_left_to_the_generic_kernel(capfd):
    """a well-formed graph gives the giant kernel nothing to refuse (a refusal is not an error -- the generic kernel then decodes
    the block -- so only the tier names tell)"""
    err = capfd.readouterr().err
    assert "tier2a (giant)" in err, "the giant kernel did not run"
    assert "tier2 (generic)" not in err, [l for l in err.splitlines() if "tier2" in l][:6]


@pytest.mark.parametrize("shape", ["eu", "web", "nowindow", "minint2", "nointervals", "nointervals_dense"])
def test_every_block_through_the_giant_kernel(W, tools, oracle, every_block, capfd, shape):
    if shape == "eu": st = tools.synth_store(6000, seed=5, synth=tools.eu_like(), threads=4)
    # (round 6, fuzz seed 601: without an interval section the header ends in the wave-parallel copy-block parse; a list of EXACTLY skip_min residuals -- 8 on sparse graphs, 16 on
    #  dense ones: no index entry -- was then decoded in step from a buffer that had not been brought back to the cursor, on every scan after the index was built)
    elif shape == "nointervals": st = tools.synth_store(20000, seed=16, params=W.default_params(min_interval_length=0, window_size=20, max_ref_count=3), synth=tools.web_like(), threads=4)
    elif shape == "nointervals_dense": st = tools.synth_store(8000, seed=17, params=W.default_params(min_interval_length=0, window_size=20, max_ref_count=3), synth=tools.eu_like(), threads=4)
    elif shape == "web": st = tools.synth_store(20000, seed=6, synth=tools.web_like(), threads=4)
    elif shape == "nowindow": st = tools.synth_store(8000, seed=7, params=W.default_params(window_size=0, max_ref_count=0), synth=tools.web_like(), threads=4)
    else: st = tools.synth_store(6000, seed=8, params=W.default_params(min_interval_length=2, window_size=12, max_ref_count=6), synth=tools.eu_like(), threads=4)
    n = st.params.nodes
    res = _check(W, oracle, st, ranges=[(0, 1), (n // 3, n), (n - 1, n), (100, 100)])
    assert res[0]["slow_blocks"] > 0
    _none_left_to_the_generic_kernel(capfd)


def test_cnr2000_through_the_giant_kernel(W, cnr_csr, every_block, capfd):
    from conftest import CNR
    g = W.BVGraph.load(CNR)
    gdeg, gsucc = cnr_csr
    r = g.scan()
    assert r["arcs"] == gsucc.size and r["slow_blocks"] > 0
    deg, succ = g.decode_range(0, g.num_nodes())
    assert np.array_equal(deg, gdeg) and np.array_equal(succ, gsucc)
    g.close()
    _none_left_to_the_generic_kernel(capfd)


def _giant_graph(rng, n, giants, deg):
    """ordinary nodes (about 10 successors nearby) + pairs of adjacent giants, the second copying most of the first"""
    d = rng.poisson(10, n).astype(np.int64)
    gpos = np.sort(rng.choice(np.arange(100, n - 100, 7), giants, replace=False))
    for i in range(0, giants - 1, 2): gpos[i + 1] = gpos[i] + 1
    lists, prev = {}, None
    for i, gx in enumerate(gpos):
        if i % 2 == 1:
            keep = prev[rng.random(prev.size) < 0.8]
            l = np.union1d(keep, rng.choice(n, deg // 5, replace=False))
        else:
            l = np.sort(rng.choice(n, deg, replace=False))
            # a few long runs of consecutive successors: intervals inside a giant
            s0 = int(rng.integers(0, n - 3000)); l = np.union1d(l, np.arange(s0, s0 + 2500))
        lists[int(gx)] = l.astype(np.int64); prev = l
    for gx in lists: d[gx] = lists[gx].size
    off = np.zeros(n + 1, np.int64); np.cumsum(d, out=off[1:])
    adj = np.empty(off[-1], np.int64)
    for x in range(n):
        if x in lists: adj[off[x]:off[x + 1]] = lists[x]
        elif d[x]: adj[off[x]:off[x + 1]] = np.sort(rng.choice(np.arange(max(0, x - 3000), min(n, x + 3000)), d[x], replace=False))
    return off.astype(np.uint64), adj


@pytest.mark.parametrize("noskip", [False, True])
def test_graph_with_giant_lists(W, tools, oracle, monkeypatch, noskip):
    if noskip: monkeypatch.setenv("BVG_NOSKIP", "1")
    rng = np.random.default_rng(3)
    off, adj = _giant_graph(rng, 60000, 6, 25000)
    st = tools.store((off, adj), threads=4)
    n = st.params.nodes
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = _oracle_graph(oracle, st)
    o = og.scan()
    for it in range(3):
        r = g.scan()
        assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"]), it
    assert r["slow_blocks"] >= 3
    deg, succ = g.decode_range(0, n)
    assert np.array_equal(deg, np.diff(off.astype(np.int64))) and np.array_equal(succ, adj)
    # random access to the giants themselves and their neighbours
    big = np.flatnonzero(np.diff(off.astype(np.int64)) > 20000)
    nodes = np.concatenate([big, big + 1, big - 1, rng.integers(0, n, 50)]).astype(np.int64)
    bdeg, bsucc = g.successors_batch(nodes)
    ref = np.concatenate([adj[int(off[x]):int(off[x + 1])] for x in nodes])
    assert np.array_equal(bdeg, np.diff(off.astype(np.int64))[nodes]) and np.array_equal(bsucc, ref)
    g.close()


def test_lists_with_more_index_tasks_than_threads(W, tools, oracle):
    """lists of 120 000 successors, every second one copying 80 % of the one before at random: ~40 000 copy blocks and ~100 000
    residuals per list, i.e. more than 1 024 index tasks per section (the task loops of the giant kernel go round more than once)"""
    rng = np.random.default_rng(8)
    off, adj = _giant_graph(rng, 300000, 4, 120000)
    st = tools.store((off, adj), threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    o = _oracle_graph(oracle, st).scan()
    for it in range(3):
        r = g.scan()
        assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"]), it
    big = np.flatnonzero(np.diff(off.astype(np.int64)) > 100000)
    bdeg, bsucc = g.successors_batch(big.astype(np.int64))
    assert np.array_equal(bsucc, np.concatenate([adj[int(off[x]):int(off[x + 1])] for x in big]))
    g.close()


def test_giant_kernel_is_what_ran(W, tools, oracle, monkeypatch, capfd):
    """BVG_DEBUG names the tiers: the giant blocks are decoded by tier 2a, none is left to the generic kernel."""
    monkeypatch.setenv("BVG_DEBUG", "1")
    rng = np.random.default_rng(4)
    off, adj = _giant_graph(rng, 40000, 4, 30000)
    st = tools.store((off, adj), threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    g.scan(); g.scan()
    g.close()
    errtxt = capfd.readouterr().err
    last = [l for l in errtxt.splitlines() if "tiers concurrent" in l][-1]
    assert " 0 generic blocks" in last and " 0 giant" not in last, last
