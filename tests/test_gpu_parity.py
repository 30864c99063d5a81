"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle and the reference's golden."""
import numpy as np
import pytest

from conftest import CNR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cnr_gpu(W):
    g = W.BVGraph.load(CNR)
    yield g
    g.close()


def _oracle_graph(O, st):
    return O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)


def test_cnr2000_full_decode_matches_reference_golden(cnr_gpu, cnr_csr):
    """BVGraphTest.testLarge (test/.../BVGraphTest.java:105-123) replayed on the HIP path."""
    deg, succ = cnr_gpu.decode_range(0, cnr_gpu.num_nodes())
    gdeg, gsucc = cnr_csr
    assert np.array_equal(deg, gdeg)
    assert np.array_equal(succ, gsucc)


def test_cnr2000_scan_checksum_matches_oracle(cnr_gpu, oracle):
    og = oracle.Graph.load(CNR)
    r = cnr_gpu.scan()
    o = og.scan()
    assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])
    assert r["arcs"] == 3216152 and r["graph_bytes"] == 1430488


def test_cnr2000_outdegrees_and_random_access(cnr_gpu, cnr_golden):
    deg = cnr_gpu.outdegrees()
    assert np.array_equal(deg, np.array([len(a) for a in cnr_golden], dtype=np.int32))
    rng = np.random.default_rng(0)
    for x in list(rng.integers(0, cnr_gpu.num_nodes(), 40)) + [0, cnr_gpu.num_nodes() - 1]:
        it = cnr_gpu.successors(int(x))
        got = list(it)
        assert got == cnr_golden[x].tolist()
        assert it.next_long() == -1          # -1 forever after the end (WebGraphTestCase.java:125)


def test_cnr2000_random_access_of_every_node(cnr_gpu, cnr_csr):
    """BVGraphTest.java:112-121 (testLarge) checks successors(x) of EVERY node of the fixture, in random order: here the whole node
    set as one frontier of bvg_successors_batch (each request decoded with its own reference chain, BVG:1084), in a random
    permutation, against the reference's golden."""
    gdeg, gsucc = cnr_csr
    n = cnr_gpu.num_nodes()
    cum = np.concatenate([[0], np.cumsum(gdeg, dtype=np.int64)])
    perm = np.random.default_rng(5).permutation(n).astype(np.int64)
    for part in np.array_split(perm, 4):
        d, s = cnr_gpu.successors_batch(part)
        assert np.array_equal(d, gdeg[part])
        exp = np.concatenate([gsucc[cum[x]:cum[x + 1]] for x in part])
        assert np.array_equal(s, exp)


@pytest.mark.parametrize("frm,to", [(0, 1), (5, 77), (1000, 5000), (325000, 325557), (123456, 123457), (64, 64)])
def test_cnr2000_subranges(cnr_gpu, cnr_golden, frm, to):
    deg, succ = cnr_gpu.decode_range(frm, to)
    exp = cnr_golden[frm:to]
    assert deg.tolist() == [len(a) for a in exp]
    assert np.array_equal(succ, np.concatenate(exp) if exp and sum(len(a) for a in exp) else np.empty(0, np.int64))


@pytest.mark.parametrize("block_bits", [2048, 8192, 262144])
def test_block_size_does_not_change_results(W, cnr_csr, oracle, block_bits):
    g = W.BVGraph.load(CNR)
    g.set_tuning(block_bits=block_bits)
    deg, succ = g.decode_range(0, g.num_nodes())
    assert np.array_equal(deg, cnr_csr[0]) and np.array_equal(succ, cnr_csr[1])
    r = g.scan()
    assert r["chk"] == oracle.Graph.load(CNR).scan()["chk"]


def test_wide_and_slow_paths_are_bit_exact(W, cnr_csr):
    for kw in ({"force_wide": True}, {"force_slow": True}, {"force_wide": True, "force_slow": True}, {"stream": True}, {"stream": True, "force_wide": True}, {"legacy": True}):
        g = W.BVGraph.load(CNR)
        g.set_tuning(**kw)
        deg, succ = g.decode_range(0, 60000)
        n = int(deg.sum())
        assert np.array_equal(deg, cnr_csr[0][:60000]) and np.array_equal(succ, cnr_csr[1][:n]), kw


@pytest.mark.parametrize("params", [
    dict(), dict(window_size=0, max_ref_count=0, min_interval_length=0), dict(window_size=1, max_ref_count=1, min_interval_length=2),
    dict(window_size=3, max_ref_count=10), dict(window_size=16, max_ref_count=2, min_interval_length=0), dict(zeta_k=1), dict(zeta_k=5),
    dict(outdegree_coding=1, block_coding=1, residual_coding=1, reference_coding=1, block_count_coding=1),      # all delta
    dict(residual_coding=2, reference_coding=2, block_count_coding=5, block_coding=5),                         # gamma residuals/refs, unary blocks
    dict(residual_coding=7), dict(residual_coding=3, zeta_k=5),                                                # nibble, Golomb(b=5)
])
def test_synthetic_graphs_match_oracle(W, tools, oracle, params):
    st = tools.synth_store(30000, seed=3, params=W.default_params(**params), threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = _oracle_graph(oracle, st)
    deg, succ = g.decode_range(0, g.num_nodes())
    odeg, osucc = og.decode_range(0, g.num_nodes())
    assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc)
    r, o = g.scan(), og.scan()
    assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])


def test_randomised_parameters_and_shapes(W, tools, oracle):
    """Randomised sweep: window / maxRef / minInterval / zeta_k / codings x graph shape, full materialise + scan
    parity against the oracle (the reference's testCompression idea, BVGraphTest.java:52-103, at random)."""
    rng = np.random.default_rng(2026)
    for trial in range(40):
        kw = dict(window_size=int(rng.choice([0, 1, 2, 7, 16, 64])), max_ref_count=int(rng.choice([0, 1, 3, 1000])),
                  min_interval_length=int(rng.choice([0, 1, 2, 4, 9])), zeta_k=int(rng.choice([1, 2, 3, 4, 7])))
        if rng.random() < 0.4:
            kw.update(outdegree_coding=int(rng.choice([1, 2])), block_coding=int(rng.choice([1, 2, 5])), residual_coding=int(rng.choice([1, 2, 3, 6, 7])),
                      reference_coding=int(rng.choice([1, 2, 5])), block_count_coding=int(rng.choice([1, 2, 5])))
            if kw["residual_coding"] == 3:
                kw["zeta_k"] = int(rng.choice([1, 3, 5, 8]))
        n = int(rng.choice([1, 2, 63, 64, 65, 500, 3000]))
        synth = tools.web_like(mean_deg=float(rng.choice([2, 10, 60])), p_copy=float(rng.choice([0.0, 0.5, 0.95])), p_empty=float(rng.choice([0.0, 0.3, 0.9])),
                               p_interval=float(rng.choice([0.0, 0.5])), max_deg=int(rng.choice([5, 300, 5000])), window=int(rng.choice([1, 7, 30])))
        st = tools.synth_store(n, seed=int(rng.integers(1 << 30)), params=W.default_params(**kw), synth=synth, chunk_nodes=int(rng.choice([64, 1 << 16])), threads=2)
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
        og = _oracle_graph(oracle, st)
        deg, succ = g.decode_range(0, n)
        odeg, osucc = og.decode_range(0, n)
        assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc), (trial, kw, n)
        r, o = g.scan(), og.scan()
        assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"]), (trial, kw, n)
        g.close()


def test_degenerate_graphs(W, tools, oracle):
    for lists in ([], [[]], [[0]], [[] for _ in range(200)], [list(range(0, 5000))], [[1], [0]], [list(range(64))] * 70):
        st = tools.store(lists, W.default_params())
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
        deg, succ = g.decode_range(0, len(lists))
        assert deg.tolist() == [len(l) for l in lists]
        assert succ.tolist() == [v for l in lists for v in l]
        r = g.scan()
        assert r["arcs"] == sum(map(len, lists)) and r["nodes"] == len(lists)
        if lists:
            og = _oracle_graph(oracle, st)
            assert r["chk"] == og.scan()["chk"]


def test_cnr2000_transpose(cnr_gpu, cnr_csr):
    """The golden graph transposed on the device against numpy on the golden adjacency."""
    gdeg, gsucc = cnr_csr
    n = len(gdeg)
    toff, tsucc = cnr_gpu.transpose()
    src = np.repeat(np.arange(n, dtype=np.int64), gdeg)
    order = np.argsort(gsucc, kind="stable")
    assert np.array_equal(toff, np.concatenate([[0], np.cumsum(np.bincount(gsucc, minlength=n))]).astype(np.uint64))
    assert np.array_equal(tsucc, src[order])


def test_cnr2000_symmetrize(cnr_gpu, cnr_csr):
    """The golden graph symmetrised on the device against numpy on the golden adjacency."""
    gdeg, gsucc = cnr_csr
    n = len(gdeg)
    soff, ssucc = cnr_gpu.symmetrize()
    src = np.repeat(np.arange(n, dtype=np.int64), gdeg)
    keys = np.unique(np.concatenate([src * n + gsucc, gsucc * n + src]))
    assert np.array_equal(soff, np.concatenate([[0], np.cumsum(np.bincount(keys // n, minlength=n))]).astype(np.uint64))
    assert np.array_equal(ssucc, keys % n)


def test_cnr2000_tiled_on_the_device_matches_the_golden_across_the_seams(W, oracle, cnr_golden, cnr_csr):
    """bench.py --shape cnr: the reference's fixture repeated on the device (bvg_mosaic of ONE base; BV records are translation
    invariant).  Tile j must be the golden (BVGraphTest.testLarge's expected lists) shifted by j * n -- also for the nodes on either
    side of every seam, whose block, halo and stream window straddle two copies of the stream -- and the scan's checksum must be the
    oracle's, tile by tile and across seams."""
    import os
    from conftest import CNR
    base = W.BVGraph.load(CNR)
    n = base.num_nodes()
    copies = 5
    g = W.mosaic([base], copies)
    assert g.num_nodes() == copies * n and g.num_arcs() == copies * base.num_arcs()
    deg, succ = cnr_csr
    cum = np.concatenate([[0], np.cumsum(deg, dtype=np.int64)])
    og = oracle.Graph.load(CNR)
    for _ in range(2):                                                  # (the second scan runs indexed)
        r = g.scan()
        assert r["arcs"] == copies * int(cum[-1]) and r["nodes"] == copies * n
    whole = 0
    for j in range(copies):
        o = og.scan(0, n, node_base=j * n)
        t = g.scan(j * n, (j + 1) * n)
        assert (t["arcs"], t["chk"]) == (o["arcs"], o["chk"]), "tile %d" % j
        whole = (whole + o["chk"]) & 0xFFFFFFFFFFFFFFFF
    assert r["chk"] == whole
    K = 3000
    for j in range(1, copies):                                          # K nodes on either side of seam j
        d, s = g.decode_range(j * n - K, j * n + K)
        assert np.array_equal(d, np.concatenate([deg[n - K:], deg[:K]]))
        want = np.concatenate([succ[cum[n - K]:] + (j - 1) * n, succ[:cum[K]] + j * n])
        assert np.array_equal(s, want), "seam %d" % j
        oa = og.scan(n - K, n, node_base=(j - 1) * n); ob = og.scan(0, K, node_base=j * n)
        sc = g.scan(j * n - K, j * n + K)
        assert sc["arcs"] == oa["arcs"] + ob["arcs"] and sc["chk"] == (oa["chk"] + ob["chk"]) & 0xFFFFFFFFFFFFFFFF
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.integers(0, copies * n, 2000), np.array([j * n + k for j in range(1, copies) for k in (-1, 0, 1, 7)])]).astype(np.int64)
    bd, bs = g.successors_batch(xs)
    exp = [succ[cum[x % n]:cum[x % n + 1]] + (x // n) * n for x in xs.tolist()]
    assert np.array_equal(bs, np.concatenate(exp)) and np.array_equal(bd, np.array([len(e) for e in exp], dtype=np.int32))
    g.close(); base.close()
