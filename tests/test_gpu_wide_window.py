"""Window sizes above 64 (BVGraph allows any windowsize; the LDS row kernels stop at 64): every block then takes the generic
global-memory kernel with a 2048-entry node ring, block halos become plain counts, and the offsets derivation keeps a ring of
2048 outdegrees.  Lists copy from FAR references here (uniform in [1, W]), so chains really reach hundreds of nodes back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _far_copy_lists(n, w, seed):
    rng = np.random.default_rng(seed)
    lists = []
    for x in range(n):
        own = set(int(v) for v in rng.integers(0, n, size=int(rng.integers(0, 6))))
        if x > 0 and rng.random() < 0.8:
            ref = lists[x - int(rng.integers(1, min(x, w) + 1))]
            own |= set(v for v in ref if rng.random() < 0.85)
        if rng.random() < 0.2:
            a = int(rng.integers(0, n - 8)); own |= set(range(a, a + int(rng.integers(4, 9))))
        lists.append(sorted(own))
    return lists


@pytest.mark.parametrize("w,maxref", [(65, 3), (100, 3), (300, 6), (1000, 2)])
def test_wide_windows_match_oracle(W, tools, oracle, w, maxref):
    n = 6000
    lists = _far_copy_lists(n, w, seed=w)
    st = tools.store(lists, W.default_params(window_size=w, max_ref_count=maxref), threads=4)
    assert st.stats["tot_dist"] > 20 * st.stats["nodes_with_ref"], "references should reach far back"
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    r, o = g.scan(), og.scan()
    assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])
    deg, succ = g.decode_range(0, n)
    assert deg.tolist() == [len(l) for l in lists] and succ.tolist() == [v for l in lists for v in l]
    for a, b in ((0, 1), (n - 1, n), (2999, 3001), (4000, 4000), (1234, 5678)):
        d2, s2 = g.decode_range(a, b)
        assert s2.tolist() == [v for l in lists[a:b] for v in l]
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"])
    nodes = np.array([n - 1, 0, 4321, 4321, 17], dtype=np.int64)                # frontier-style random access
    bd, bs = g.successors_batch(nodes)
    assert bs.tolist() == [v for x in nodes for v in lists[int(x)]]
    toff, tsucc = g.transpose()
    src = np.repeat(np.arange(n, dtype=np.int64), deg)
    assert np.array_equal(tsucc, src[np.argsort(succ, kind="stable")])
    g.close()
    # offsets derived on the device from the bare stream (loadSequential / loadOffline)
    g2 = W.BVGraph.from_memory(st.params, st.graph, None)
    assert np.array_equal(g2.offsets(), st.offsets)
    g2.close()


def test_window_beyond_the_ring_is_refused(W, tools):
    st = tools.store([[1], [0]], W.default_params(window_size=7))
    p = st.params.clone(window_size=5000)
    with pytest.raises(W.UnsupportedOperationException):
        W.BVGraph.from_memory(p, st.graph, st.offsets)
