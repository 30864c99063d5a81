"""GPU: the strong-scaling shard path (BASELINE config 5) through the C ABI — bvg_shard_bounds / bvg_scan_shard /
bvg_scan_multi — on one device: k shards of ONE graph add up to its one-piece scan for every k and balance, the bounds follow
the rules bench.py's helper (webgraph-big_amd/shard.py) states on the host, and bvg_copy() flyweights may scan concurrently
from several host threads, also with different block sizes (each holds its plan by reference count).
Reference: ImmutableGraph.java:405-436 (splitNodeIterators), algo/HyperBall.java:748-768 (arc-balanced tasks)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def graph(W, tools, oracle):
    st = tools.synth_store(50000, seed=31, synth=tools.eu_like(mean_deg=40.0), threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    return g, og, st


@pytest.mark.parametrize("balance", ["nodes", "bits", "arcs"])
def test_shards_add_up_to_the_whole_scan(W, graph, balance):
    from webgraph_big_amd import shard as S
    g, og, st = graph
    bal = {"nodes": W.BALANCE_NODES, "bits": W.BALANCE_BITS, "arcs": W.BALANCE_ARCS}[balance]
    whole = og.scan()
    n = g.num_nodes()
    deg = g.outdegrees()
    for k in (1, 2, 3, 8):
        b = g.shard_bounds(k, bal)
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(k))
        if balance == "nodes":
            assert b.tolist() == [lo for lo, _ in S.split_nodes(n, k)] + [n]
        if balance == "arcs":
            assert b.tolist() == S.bounds_by_arcs(deg, k).tolist()
        arcs = chk = nodes = gbytes = 0
        for r in range(k):
            # the same helper bench.py calls on every rank, without the collective
            res, a, c = S.sharded_scan(lambda lo, hi: g.scan(lo, hi), b, r, reduce=False)
            res2 = g.scan_shard(k, r, bal)
            assert (res2["from"], res2["to"]) == (b[r], b[r + 1]) and (res2["arcs"], res2["chk"]) == (a, c)
            o = og.scan(int(b[r]), int(b[r + 1]))
            assert (a, c) == (o["arcs"], o["chk"]), (k, r)
            arcs += a; chk = (chk + c) % (1 << 64); nodes += res["nodes"]; gbytes += res["graph_bytes"]
        assert (nodes, arcs, chk) == (whole["nodes"], whole["arcs"], whole["chk"]), k
        assert abs(gbytes - len(st.graph)) <= k                     # shards share at most one byte at each seam
        if balance == "arcs" and k > 1:
            per = [int(deg[b[r]:b[r + 1]].sum()) for r in range(k)]
            assert max(per) - min(per) <= 2 * int(deg.max()) + 2


def test_scan_multi_over_flyweights(W, graph):
    g, og, st = graph
    whole = og.scan()
    for k in (1, 2, 4):
        hs = [g.copy() for _ in range(k)]
        tot, per = W.scan_multi(hs, W.BALANCE_ARCS)
        assert (tot["nodes"], tot["arcs"], tot["chk"]) == (whole["nodes"], whole["arcs"], whole["chk"])
        assert len(per) == k and sum(p["arcs"] for p in per) == whole["arcs"]
        b = g.shard_bounds(k, W.BALANCE_ARCS)
        for r in range(k):
            o = og.scan(int(b[r]), int(b[r + 1]))
            assert (per[r]["arcs"], per[r]["chk"]) == (o["arcs"], o["chk"])
        for h in hs:
            h.close()


def test_copies_with_different_block_sizes_scan_concurrently(W, graph):
    """a flyweight with another block_bits used to rebuild (and free) the plan under the others' kernels."""
    g, og, st = graph
    whole = og.scan()
    want = (whole["nodes"], whole["arcs"], whole["chk"])
    hs = [g.copy() for _ in range(4)]
    for i, h in enumerate(hs):
        h.set_tuning(block_bits=[0, 16384, 65536, 24576][i])
    out = [None] * 4
    def work(i):
        res = []
        for _ in range(6):
            r = hs[i].scan()
            res.append((r["nodes"], r["arcs"], r["chk"]))
        out[i] = res
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    for i in range(4):
        assert out[i] == [want] * 6, i
    for h in hs:
        h.close()


def test_every_shard_of_an_8_way_split_scans_with_its_part_of_the_skip_index(W, tools, oracle):
    """VERDICT r2 / ADVICE r2: a shard smaller than a quarter of the graph used to scan index-less.  Now the first scan of a shard
    indexes exactly the blocks it covers: 8 shards hold 8 disjoint parts that add up to the whole index, every shard reads
    skip entries, and the whole-graph scan afterwards still agrees.  (ImmutableGraph.java:405-436, algo/HyperBall.java:748-768)"""
    from webgraph_big_amd import shard as S
    st = tools.synth_store(200000, seed=5, synth=tools.eu_like(mean_deg=80.0), threads=4)
    og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    whole = og.scan()
    ref = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    e_all, b_all = ref.build_index()
    assert e_all > 1000 and b_all >= 6 * e_all
    r_all = ref.scan()
    assert r_all["index_entries"] == e_all and (r_all["arcs"], r_all["chk"]) == (whole["arcs"], whole["chk"])
    k = 8
    bounds = ref.shard_bounds(k, W.BALANCE_ARCS)
    ent = 0; arcs = 0; chk = 0
    for r in range(k):
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)      # a replica per rank, as on the 8-GPU node
        res, a, c = S.sharded_scan(lambda lo, hi: g.scan(lo, hi), bounds, r, reduce=False)
        assert res["index_entries"] > 0, "shard %d scanned without skip entries" % r
        e_r, _ = g.build_index(int(bounds[r]), int(bounds[r + 1]))      # covered already: reports what the replica holds
        assert 0 < e_r < e_all / 4, (r, e_r, e_all)                     # its own part, not the whole index
        o = og.scan(int(bounds[r]), int(bounds[r + 1]))
        assert (a, c) == (o["arcs"], o["chk"])
        ent += res["index_entries"]; arcs += a; chk = (chk + c) % (1 << 64)
        if r == 3:                                                      # a scan outside the shard re-indexes the whole graph, once
            rr = g.scan()
            assert (rr["arcs"], rr["chk"]) == (whole["arcs"], whole["chk"]) and rr["index_entries"] == e_all
            assert g.scan(int(bounds[r]), int(bounds[r + 1]))["chk"] == c
        g.close()
    assert (arcs, chk) == (whole["arcs"], whole["chk"])
    assert 0 <= ent - e_all <= 512 * k                                  # a block at a seam is indexed by both neighbours
    ref.close()


def test_scan_multi_rejects_one_handle_twice(W, graph):
    g, og, st = graph
    with pytest.raises(W.IllegalArgumentException):
        W.scan_multi([g, g])
