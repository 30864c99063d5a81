"""GPU: BVGraph.store on the device (bvg_store, csrc/bvg_encode.hip; SURVEY §8(f) rank 4, second half).
Reference: BVGraph.java:1595-1618 (intervalize), :1977-2159 (diffComp), :2216-2327 (reference selection), :2404-2457 (per-thread ranges).
The bar is byte-exactness: the reference's own fixture cnr-2000.graph / .offsets must be regenerated from the text golden, and on
synthetic graphs the device output must equal the CPU tooling's (which is itself held to the fixture, tests/test_store.py) for
every coding, window, chunking and degenerate shape; what was written must decode back to the adjacency on the HIP path."""
import numpy as np
import pytest

from conftest import CNR

pytestmark = pytest.mark.gpu


def _csr(lists):
    off = np.zeros(len(lists) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(l) for l in lists], dtype=np.uint64) if len(lists) else 0
    adj = np.concatenate([np.asarray(l, dtype=np.int64) for l in lists]) if len(lists) and off[-1] else np.empty(0, np.int64)
    return off, adj


def test_cnr2000_fixture_is_regenerated_byte_for_byte(W, tools, cnr_golden):
    p = W.parse_properties(open(CNR + ".properties").read())
    graph, offsets = W.store(_csr(cnr_golden), p)
    assert graph.tobytes() == open(CNR + ".graph", "rb").read()
    assert tools.encode_offsets(offsets, p.offset_coding).tobytes() == open(CNR + ".offsets", "rb").read()


@pytest.mark.parametrize("kw", [
    dict(),
    dict(window_size=0, max_ref_count=0, min_interval_length=0),
    dict(window_size=1, max_ref_count=1, min_interval_length=2),
    dict(window_size=16, max_ref_count=-1, min_interval_length=3, zeta_k=5),
    dict(window_size=7, max_ref_count=3, min_interval_length=0),
    dict(outdegree_coding=1, block_coding=1, residual_coding=1, reference_coding=1, block_count_coding=1),
    dict(residual_coding=2, reference_coding=2, block_count_coding=5, block_coding=5),
    dict(residual_coding=7), dict(residual_coding=3, zeta_k=4),
])
@pytest.mark.parametrize("chunk", [0, 1000])
def test_device_store_equals_cpu_tooling(W, tools, kw, chunk):
    off, adj = tools.synth_adjacency(7000, seed=23, synth=tools.eu_like(mean_deg=40.0), chunk_nodes=1 << 16)
    p = W.default_params(**kw)
    want = tools.store((off, adj), p, chunk_nodes=chunk)
    graph, offsets = W.store((off, adj), p, chunk_nodes=chunk)
    assert np.array_equal(offsets, want.offsets), kw
    assert graph.tobytes() == want.graph.tobytes(), kw


def test_round_trip_through_the_hip_decoder(W, tools, oracle):
    off, adj = tools.synth_adjacency(20000, seed=29, synth=tools.web_like(), chunk_nodes=1 << 16)
    p = W.default_params().clone(nodes=20000, arcs=len(adj))
    graph, offsets = W.store((off, adj), p)
    g = W.BVGraph.from_memory(p, graph, offsets)
    deg, succ = g.decode_range(0, 20000)
    assert np.array_equal(deg, np.diff(off).astype(np.int32)) and np.array_equal(succ, adj)
    g.close()


def test_degenerate_adjacencies(W, tools):
    for lists in ([], [[]], [[0]], [[], [], []], [[1, 2], [0, 1, 2], [0, 1, 2]], [list(range(50))] * 50 + [[]] * 3 + [[5, 49]]):
        lists = [l for l in lists]
        n = len(lists)
        lists = [[v for v in l if v < max(n, 1)] for l in lists]
        p = W.default_params()
        want = tools.store(lists, p) if n else None
        graph, offsets = W.store(lists, p)
        if n:
            assert np.array_equal(offsets, want.offsets) and graph.tobytes() == want.graph.tobytes(), lists
        else:
            assert len(graph) == 0 and offsets.tolist() == [0]
    with pytest.raises(W.IllegalArgumentException):
        W.store([[1, 1], [0]], W.default_params())                  # duplicate successor (BVG:2141)
    with pytest.raises(W.IllegalArgumentException):
        W.store([[0, 5]], W.default_params())                       # successor outside the graph
