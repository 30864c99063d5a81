"""GPU: a graph with MORE THAN 2^31 NODES (the "big" of webgraph-big: node ids and successors are longs, BVGraph.java:654-760;
the reference's own slow test builds such graphs, slow/.../BVGraphSlowTest.java:87-96).  A sparse base graph is tiled on the
device past 2^31 nodes, so the 64-bit successor kernels run on real 64-bit values; checked by size-independent properties:
arc count, tiles on either side of node 2^31 against the CPU oracle with the matching node base, additivity of range scans,
and materialised windows across the 2^31 boundary."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,wide_from_2_31", [(31, False), (31, True), (32, False)])
def test_scan_and_decode_past_2_to_31_nodes(W, tools, oracle, monkeypatch, bits, wide_from_2_31):
    """bits = 31: ids and successors in [2^31, 2^32) -- still the 32-bit successor kernels (they hold every id below 2^32 - 1; the
    reference splits at 2^31 only because Java ints are signed); bits = 32: more than 2^32 nodes, the 64-bit kernels on real 64-bit values."""
    if wide_from_2_31: monkeypatch.setenv("BVG_WIDE_FROM_2_31", "1")           # round 1's switch: the 64-bit kernels on ids in [2^31, 2^32)
    n0 = 1 << 19
    st = tools.synth_store(n0, seed=77, synth=tools.web_like(mean_deg=3.0, p_empty=0.5, max_deg=200), threads=4)
    tiles = (1 << bits) // n0 + 3
    base = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    try:
        g = base.tile(tiles)                                  # ~18 GB of offsets for 2^31 nodes
    except MemoryError:
        pytest.skip("needs ~25 GB (2^31) / ~50 GB (2^32) of HBM")
    n = g.num_nodes()
    assert n > (1 << bits) and g.num_arcs() == st.stats["arcs"] * tiles
    og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    whole = g.scan()
    assert whole["nodes"] == n and whole["arcs"] == st.stats["arcs"] * tiles
    jb = (1 << bits) // n0
    for j in (0, jb - 1, jb, tiles - 1):                       # tiles below, across and above node 2^31
        r = g.scan(j * n0, (j + 1) * n0)
        o = og.scan(0, n0, node_base=j * n0, threads=4)
        assert (r["arcs"], r["chk"]) == (o["arcs"], o["chk"]), j
    # additivity: four range scans add up to the whole (mod 2^64)
    cuts = [0, n // 3 + 17, (1 << bits) - 5, (1 << bits) + 12345, n]
    parts = [g.scan(cuts[i], cuts[i + 1]) for i in range(4)]
    assert sum(p["arcs"] for p in parts) == whole["arcs"] and sum(p["chk"] for p in parts) % (1 << 64) == whole["chk"]
    # materialised successors across the boundary: ids and successors beyond 2^31, bit-exact against the shifted base lists
    lo, hi = (1 << bits) - 300, (1 << bits) + 300
    deg, succ = g.decode_range(lo, hi)
    odeg, osucc = og.decode_range(n0 - 300, n0)
    odeg2, osucc2 = og.decode_range(0, 300)
    want = np.concatenate([osucc + (jb - 1) * n0, osucc2 + jb * n0])
    assert np.array_equal(deg, np.concatenate([odeg, odeg2])) and np.array_equal(succ, want)
    assert succ.max() > (1 << bits)
    sb = g.successors_batch(np.array([(1 << bits) + 7, 5, n - 1], dtype=np.int64))
    exp = np.concatenate([og.successors(7) + jb * n0, og.successors(5), og.successors(n0 - 1) + (tiles - 1) * n0])
    assert np.array_equal(sb[1], exp)
    g.close(); base.close()
