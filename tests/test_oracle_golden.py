"""CPU: the oracle restatement pinned on the reference's own golden (BVGraphTest.testLarge,
test/it/unimi/dsi/big/webgraph/BVGraphTest.java:105-123) and on the API contract of
WebGraphTestCase.assertGraph (test/.../WebGraphTestCase.java:106-199)."""
import numpy as np
import pytest

from conftest import CNR


@pytest.fixture(scope="module")
def cnr(oracle):
    return oracle.Graph.load(CNR)


def test_properties_of_fixture(cnr):
    p = cnr.params
    assert (p.nodes, p.arcs, p.window_size, p.max_ref_count, p.min_interval_length, p.zeta_k) == (325557, 3216152, 7, 3, 3, 3)


def test_sequential_decode_equals_text_golden(cnr, cnr_csr):
    deg, succ = cnr.decode_range(0, cnr.num_nodes())
    assert np.array_equal(deg, cnr_csr[0])
    assert np.array_equal(succ, cnr_csr[1])
    assert int(deg.sum()) == 3216152 and int(deg.max()) == 2716          # SURVEY 8c invariants


def test_random_access_equals_text_golden(cnr, cnr_golden):
    rng = np.random.default_rng(1)
    for x in list(range(0, 3000)) + list(rng.integers(0, cnr.num_nodes(), 3000)):
        assert cnr.outdegree(int(x)) == len(cnr_golden[x])
        assert np.array_equal(cnr.successors(int(x)), cnr_golden[x])


def test_offsets_land_on_every_record(cnr, oracle):
    """the cursor after node x equals offsets[x+1]; the last offset is the stream length."""
    off = cnr.offsets()
    assert int(off[0]) == 0 and int(off[-1]) == 11443904 and (int(off[-1]) + 7) // 8 == 1430488
    it = cnr.node_iterator(0)
    for x in range(20000):
        assert it.next() == x
        assert it.bit_position() == int(off[x + 1])


@pytest.mark.parametrize("start", [1, 2, 7, 8, 9, 100, 4097, 325556, 325557])
def test_node_iterator_from_any_start_agrees_with_random_access(cnr, cnr_golden, start):
    """WebGraphTestCase.java:151-180 (the warm-up of BVGraph.java:1135-1146)."""
    it = cnr.node_iterator(start)
    for x in range(start, min(start + 300, cnr.num_nodes())):
        assert it.next() == x
        assert np.array_equal(it.successors(), cnr_golden[x])
    if start == cnr.num_nodes():
        assert not it.has_next()


def test_scan_is_split_invariant(cnr):
    a = cnr.scan()
    b = cnr.scan(threads=5)
    assert a == b and a["arcs"] == 3216152
    s1, s2 = cnr.scan(0, 100000), cnr.scan(100000, 325557)
    assert (s1["chk"] + s2["chk"]) % (1 << 64) == a["chk"] and s1["arcs"] + s2["arcs"] == a["arcs"]


def test_errors(cnr, oracle):
    with pytest.raises(oracle.OracleError) as e:
        cnr.outdegree(325557)
    assert e.value.code == -1                                              # IllegalArgumentException, BVGraph.java:823
    with pytest.raises(oracle.OracleError):
        cnr.node_iterator(-1)
    with pytest.raises(oracle.OracleError) as e:
        oracle.parse_properties("graphclass=foo.Bar\nnodes=1\n")
    assert e.value.code == -4                                              # IOException, BVGraph.java:1492
    with pytest.raises(oracle.OracleError):
        oracle.parse_properties("graphclass=it.unimi.dsi.big.webgraph.BVGraph\nnodes=1\nversion=1\n")
    with pytest.raises(oracle.OracleError):
        oracle.parse_properties("graphclass=it.unimi.dsi.big.webgraph.BVGraph\nnodes=1\ncompressionflags=RESIDUALS_UNARY\n")
    p = oracle.parse_properties("graphclass = class it.unimi.dsi.webgraph.BVGraph\nnodes=3\narcs=2\ncompressionflags=RESIDUALS_DELTA | OFFSETS_DELTA\n")
    assert p.residual_coding == 1 and p.offset_coding == 1 and p.window_size == 7     # standard class name remapped, BVGraph.java:1491


def test_mix_definition(oracle):
    """The checksum arithmetic written out independently (include/bvgraph_hip.h)."""
    M = (1 << 64) - 1

    def mix(x, y):
        m = 0xFFFFFFFF
        h = ((x & m) * 0x9E3779B1 + (x >> 32) * 0x85EBCA77) & m
        h ^= h >> 15; h = (h * 0x2C1B3C6D) & m; h ^= h >> 12
        k1 = h | 1
        k0 = (h * 0x297A2D39) & m; k0 ^= k0 >> 15
        return (k1 * y + k0) & M

    for x, y in [(0, 0), (1, 2), (325556, 17), (1 << 40, (1 << 35) + 5), (M, M)]:
        assert oracle.mix(x, y) == mix(x, y)
