"""CPU: the tooling encoder (webgraph-big_amd/tools/bvg_store.cpp) is held to the reference's bytes:
re-storing the text golden with the fixture's parameters reproduces cnr-2000.graph / .offsets exactly
(SURVEY section 7 step 3), and store -> oracle decode round-trips (BVGraphTest.testCompression,
test/.../BVGraphTest.java:52-103)."""
import itertools

import numpy as np
import pytest

from conftest import CNR


def test_restore_reproduces_fixture_bytes(W, tools, cnr_golden):
    st = tools.store(cnr_golden, W.default_params(min_interval_length=3))
    assert st.graph.tobytes() == open(CNR + ".graph", "rb").read()
    assert st.offsets_file().tobytes() == open(CNR + ".offsets", "rb").read()
    s = st.stats
    assert (s["copied"], s["intervalised"], s["residual"], s["nodes_with_ref"]) == (2130833, 361894, 723425, 181798)
    assert round(s["tot_ref"] / 325557, 2) == 1.38 and round(s["tot_dist"] / 325557, 2) == 1.74


def _binary_tree(n, out=True):
    lists = [[] for _ in range(n)]
    for i in range(n):
        for c in (2 * i + 1, 2 * i + 2):
            if c < n:
                (lists[i] if out else lists[c]).append(c if out else i)
    return [sorted(l) for l in lists]


@pytest.mark.parametrize("w,r,mi", list(itertools.product([0, 1, 2], [0, 1, 2], [0, 1, 2, 3])))
def test_compression_roundtrip_small_trees(W, tools, oracle, w, r, mi):
    """BVGraphTest.testCompression: complete binary in/out-trees n=1..7 x window x maxRef x minInterval."""
    for n in range(1, 8):
        for out in (True, False):
            lists = _binary_tree(n, out)
            st = tools.store(lists, W.default_params(window_size=w, max_ref_count=r, min_interval_length=mi))
            assert len(st.graph) == (int(st.offsets[-1]) + 7) // 8                       # file length == ceil(bits/8), :68-74
            assert st.stats["copied"] + st.stats["intervalised"] + st.stats["residual"] == sum(map(len, lists))   # :76
            og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
            for x in range(n):
                assert og.successors(x).tolist() == lists[x]
            it = og.node_iterator(0)
            for x in range(n):
                assert it.next() == x and it.successors().tolist() == lists[x]


@pytest.mark.parametrize("kw", [dict(), dict(residual_coding=1), dict(residual_coding=2), dict(residual_coding=7), dict(residual_coding=3, zeta_k=4),
                                dict(outdegree_coding=1, block_coding=1, reference_coding=1, block_count_coding=1, offset_coding=1),
                                dict(reference_coding=2, block_count_coding=5, block_coding=5), dict(zeta_k=1), dict(zeta_k=7),
                                dict(window_size=0, max_ref_count=0, min_interval_length=0), dict(window_size=30, max_ref_count=-1)])
def test_synthetic_roundtrip_all_codings(W, tools, oracle, kw):
    n = 4000
    p = W.default_params(**kw)
    st = tools.synth_store(n, seed=11, params=p, chunk_nodes=1024, threads=3)
    off, adj = tools.synth_adjacency(n, seed=11, chunk_nodes=1024)
    og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    deg, succ = og.decode_range(0, n)
    assert np.array_equal(deg, np.diff(off).astype(np.int32)) and np.array_equal(succ, adj)
    # offsets file round-trip with the chosen offset coding
    ob = st.offsets_file().tobytes()
    assert np.array_equal(oracle.decode_offsets(ob, n, st.params.offset_coding), st.offsets)
    # properties text round-trip
    if kw.get("block_coding") != 5:      # BVGraph declares no BLOCKS_UNARY constant (BVGraph.java:482-485): not expressible in .properties
        p2 = oracle.parse_properties(st.properties_text())
        assert p2.as_dict() == {**st.params.as_dict()}
    else:
        with pytest.raises(oracle.OracleError):
            oracle.parse_properties(st.properties_text())


def test_store_is_thread_count_invariant(W, tools):
    a = tools.synth_store(30000, seed=5, chunk_nodes=4096, threads=1)
    b = tools.synth_store(30000, seed=5, chunk_nodes=4096, threads=7)
    assert a.graph.tobytes() == b.graph.tobytes() and np.array_equal(a.offsets, b.offsets)


def test_hand_built_branch_graphs(W, tools, oracle):
    """One node per decoder branch of BVGraph.java:1003-1064."""
    lists = [
        [],                                   # d = 0
        [5, 9, 13, 20],                       # ref = 0, residuals only
        [5, 9, 13, 20],                       # ref = 1, zero blocks (copy everything)
        [5, 9, 13],                           # ref, odd block count (tail dropped)
        [9, 13, 20],                          # first block 0
        [9, 13, 20, 30, 31, 32, 33, 34],      # copy + interval
        [0, 1, 2, 3, 4, 5, 6],                # interval with negative first left (relative to x = 6)
        [2, 40, 41, 42, 43, 50],              # negative first residual, interval in the middle
        list(range(100, 400)),                # long interval
        [3] + list(range(100, 400)) + [999],  # copy of a long list + extras on both sides
    ]
    for w, r, mi in [(7, 3, 4), (7, 3, 2), (1, 1, 0), (0, 0, 0), (3, 1000, 3)]:
        st = tools.store(lists, W.default_params(window_size=w, max_ref_count=r, min_interval_length=mi))
        og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
        for x, l in enumerate(lists):
            assert og.successors(x).tolist() == l
        deg, succ = og.decode_range(0, len(lists))
        assert succ.tolist() == [v for l in lists for v in l]
