"""GPU: the lean scan kernel (csrc/bvg_scan.hip) — the steady-state tier 0 of an indexed scan — against the CPU oracle.

It decodes only blocks that the index-building pass of the row kernel has VALIDATED (bvg_build_index / the first scan), sums every
residual when it is decoded, never materialises a list that no later node copies and adds the kept elements of referenced lists as
runs (MaskedLongIterator.java:73-100, LongIntervalSequenceIterator.java:57-78, BVGraph.java:1062-1090).  Here: it really runs
(`lean_blocks`), it agrees with the oracle on every graph shape, sub-range, node base and pool size, and blocks holding records
whose streams overlap (which MergedLongIterator.java:85-89 would de-duplicate) are never given to it."""
import numpy as np
import pytest

from bvrecords import Record, assemble

pytestmark = pytest.mark.gpu

KNOBS = ("BVG_EMIT", "BVG_DBG", "BVG_NOSKIP", "BVG_SCANK", "BVG_SCAN_POOL", "BVG_SCAN_WAVES", "BVG_GIANT", "BVG_WIDE_HALF", "BVG_NO_LISTCUT", "BVG_MAT_LEAN", "BVG_NO_D2", "BVG_DEBUG")


@pytest.fixture(autouse=True)
def clean_env(monkeypatch):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("BVG_EMIT", "1")                                # the task variant validates; sparse graphs would take the pipelined one


def _og(O, st):
    return O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)


def _same(r, o):
    return (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])


@pytest.mark.parametrize("shape", ["eu", "eu_dense", "web", "w0", "heavy_tail", "intervals"])
def test_scan_kernel_matches_oracle(W, tools, oracle, shape):
    kw, synth, n = {}, None, 40000
    if shape == "eu": synth = tools.eu_like()
    if shape == "eu_dense": synth = tools.eu_like(mean_deg=127.5)
    if shape == "web": synth = tools.web_like(mean_deg=30.0)
    if shape == "w0": synth, kw = tools.web_like(mean_deg=40.0), dict(window_size=0, max_ref_count=0, min_interval_length=0)
    if shape == "heavy_tail": synth, n = tools.eu_like(max_deg=30000, tail_alpha=1.6, mean_deg=40.0), 15000
    if shape == "intervals": synth, kw = tools.eu_like(p_interval=0.9, interval_len=40.0), dict(min_interval_length=2)
    st = tools.synth_store(n, seed=41, params=W.default_params(**kw), synth=synth, threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = _og(oracle, st)
    o = og.scan()
    g.build_index()
    for rnd in range(3):                                               # learned tiers settle after the first scan
        r = g.scan()
        assert _same(r, o), (shape, rnd)
    assert r["lean_blocks"] > 0 and r["index_entries"] >= 0
    if shape in ("eu", "eu_dense", "w0"):
        assert r["lean_blocks"] >= 0.6 * (r["lean_blocks"] + r["slow_blocks"]), r
    rng = np.random.default_rng(3)
    for _ in range(12):
        a, b = sorted(int(v) for v in rng.integers(0, n + 1, 2))
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), (shape, a, b)
    for base in (0xFFFFF000, (1 << 33) + 7):                           # the node base folded into the checksum key, with and without carries
        g.set_node_base(base)
        assert g.scan()["chk"] == og.scan(0, n, node_base=base)["chk"], (shape, base)
    g.set_node_base(0)
    h = g.copy()                                                       # a flyweight shares the index and the validation
    rh = h.scan(n // 3, n)
    oh = og.scan(n // 3, n)
    assert (rh["arcs"], rh["chk"]) == (oh["arcs"], oh["chk"]) and rh["lean_blocks"] > 0
    h.close(); g.close()


@pytest.mark.parametrize("pool", ["640", "1024", "3072"])
def test_scan_kernel_pool_sizes(W, tools, oracle, monkeypatch, pool):
    """Rows cut short by the pool, the scratch area or the run queue, and blocks failing over to the row kernel's tiers."""
    monkeypatch.setenv("BVG_SCAN_POOL", pool)
    st = tools.synth_store(30000, seed=43, synth=tools.eu_like(mean_deg=100.0), threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    o = _og(oracle, st).scan()
    for _ in range(3):
        r = g.scan()
        assert _same(r, o), pool
    assert r["lean_blocks"] > 0
    g.close()


def test_scan_kernel_on_the_reference_fixture(W, oracle):
    """cnr-2000 (the reference's own fixture, BVGraphTest.testLarge): sparse, long reference chains."""
    from conftest import CNR
    g = W.BVGraph.load(CNR)
    og = oracle.Graph.load(CNR)
    o = og.scan()
    g.build_index()
    for _ in range(2):
        r = g.scan()
        assert _same(r, o)
    assert r["lean_blocks"] > 0
    g.close()


def test_parameters_sweep(W, tools, oracle):
    rng = np.random.default_rng(91)
    for trial in range(14):
        kw = dict(window_size=int(rng.choice([1, 3, 7, 16, 40])), max_ref_count=int(rng.choice([1, 3, 10, 1000])),
                  min_interval_length=int(rng.choice([0, 2, 4])), zeta_k=int(rng.choice([1, 2, 3, 5])))
        n = int(rng.choice([4100, 9000, 20000]))
        synth = tools.web_like(mean_deg=float(rng.choice([8, 30, 120])), p_copy=float(rng.choice([0.3, 0.6, 0.95])), p_empty=float(rng.choice([0.0, 0.3])),
                               p_interval=float(rng.choice([0.0, 0.6])), max_deg=int(rng.choice([50, 2000, 9000])), window=int(rng.choice([1, 7])))
        st = tools.synth_store(n, seed=int(rng.integers(1 << 30)), params=W.default_params(**kw), synth=synth, chunk_nodes=1 << 12, threads=2)
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
        og = _og(oracle, st)
        o = og.scan()
        for _ in range(2):
            r = g.scan()
            assert _same(r, o), (trial, kw, n)
        assert r["lean_blocks"] > 0, (trial, kw, n)
        a, b = sorted(int(v) for v in rng.integers(0, n + 1, 2))
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), (trial, kw, n, a, b)
        g.close()


def test_blocks_with_overlapping_streams_never_reach_the_scan_kernel(W, tools, oracle):
    """Odd records (a residual repeating a copied element: MergedLongIterator.java:85-89 emits it once) sprinkled over an ordinary
    graph of more than 4096 nodes: the validating pass refuses their blocks, the lean kernel takes the others, the result is exact."""
    n = 12000
    st = tools.synth_store(n, seed=5, synth=tools.web_like(mean_deg=20.0), threads=2)
    og0 = _og(oracle, st)
    deg0, succ0 = og0.decode_range(0, n)
    cum = np.concatenate([[0], np.cumsum(deg0)])
    recs, prev, odd = [], None, 0
    for x in range(n):
        l = succ0[cum[x]:cum[x + 1]].tolist()
        if x % 397 == 50 and prev is not None and len(prev) >= 4:
            recs.append(Record(d=5, ref=1, blocks=[3], residuals=[prev[1], prev[2] + 1 if prev[2] + 1 not in prev else prev[-1] + 7]))
            prev = None; odd += 1
        else:
            recs.append(Record(d=len(l), residuals=l)); prev = l
    gbytes, offs, lists = assemble(recs)
    p = W.default_params().clone(nodes=n, arcs=int(sum(r.d for r in recs)))
    og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), gbytes, offs)
    hg = W.BVGraph.from_memory(p, np.frombuffer(gbytes, dtype=np.uint8), offs)
    o = og.scan()
    assert odd >= 20
    for _ in range(3):
        r = hg.scan()
        assert _same(r, o)
    assert r["lean_blocks"] > 0 and r["slow_blocks"] > 0, r
    hg.close()


def test_reference_free_lists_decoded_in_place_and_copied_across_super_rows(W, oracle):
    """Hand-assembled records: every fourth node holds a long list WITHOUT reference and WITHOUT intervals (the scan kernel decodes such
    a stored list straight into its place: nothing parked, no position task), the three nodes after it copy from it with reference 1, 2
    and 3 (all kept / a prefix / two kept blocks) and add residuals of their own (BVGraph.java:1062-1090).  The records are large (up to
    ~2 000 bits), so a super-row holds a handful of them, ends wherever the window ends, and the references of the next one are read
    from memory, not from the window."""
    rng = np.random.default_rng(77)
    n = 6000
    recs, lists = [], []
    for x in range(n):
        lo, hi = max(0, x - 2500), min(n, x + 2500)
        if x % 4 == 0:
            l = sorted(int(v) for v in rng.choice(np.arange(lo, hi), size=int(rng.integers(40, 160)), replace=False))
            recs.append(Record(d=len(l), residuals=l))
        else:
            ref = x % 4
            base = lists[x - ref]
            if ref == 1: blocks, kept = [], list(base)                                   # no blocks: everything is kept (BVG:1030)
            elif ref == 2: k = len(base) // 2; blocks, kept = [k], base[:k]              # one block: a prefix
            else:
                k1, k2 = len(base) // 4, len(base) // 3
                blocks, kept = [k1, k2], base[:k1] + base[k1 + k2:]                      # keep, skip, keep the rest
            pool = np.setdiff1d(np.arange(lo, hi), np.array(kept, dtype=np.int64))
            extra = sorted(int(v) for v in rng.choice(pool, size=int(rng.integers(0, 30)), replace=False))
            l = sorted(kept + extra)
            recs.append(Record(d=len(l), ref=ref, blocks=blocks, residuals=extra))
        lists.append(l)
    gbytes, offs, expect = assemble(recs)
    assert expect == lists
    p = W.default_params().clone(nodes=n, arcs=int(sum(len(l) for l in lists)))
    og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), gbytes, offs)
    g = W.BVGraph.from_memory(p, np.frombuffer(gbytes, dtype=np.uint8), offs)
    o = og.scan()
    assert o["arcs"] == p.arcs
    for _ in range(3):
        r = g.scan()
        assert _same(r, o)
    assert r["lean_blocks"] >= 0.9 * (r["lean_blocks"] + r["slow_blocks"]), r
    for a, b in ((1, n - 1), (1234, 4321), (4001, 4003)):
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), (a, b)
    deg, succ = g.decode_range(0, n)                                           # the materialising kernels on the same stream
    assert succ.tolist() == [v for l in lists for v in l]
    g.close()


@pytest.mark.parametrize("half", [None, "2500"])
def test_wide_graphs_take_the_scan_kernel_with_block_relative_ids(W, tools, oracle, monkeypatch, half):
    """Graphs beyond 2^32 nodes (here: the 64-bit path forced on a small graph) run the same scan kernel on 32-bit lists of ids RELATIVE to
    a per-block base 2^31 below the block's first node; a block holding an id outside its 2^32-id window stays with the 64-bit row
    kernel.  With the window shrunk to 2 x 2 500 ids around each block both happen, and every result is the oracle's."""
    if half: monkeypatch.setenv("BVG_WIDE_HALF", half)
    n = 30000
    st = tools.synth_store(n, seed=47, synth=tools.eu_like(), threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    g.set_tuning(force_wide=True)
    og = _og(oracle, st)
    o = og.scan()
    g.build_index()
    for _ in range(3):
        r = g.scan()
        assert _same(r, o), half
    assert r["lean_blocks"] > 0, r
    if half: assert r["slow_blocks"] > 0, r                             # blocks with far successors
    else: assert r["lean_blocks"] >= 0.6 * (r["lean_blocks"] + r["slow_blocks"]), r
    for base in ((1 << 40) + 1, 0xFFFFF000):
        g.set_node_base(base)
        assert g.scan()["chk"] == og.scan(0, n, node_base=base)["chk"], (half, base)
    g.set_node_base(0)
    rng = np.random.default_rng(5)
    for _ in range(6):
        a, b = sorted(int(v) for v in rng.integers(0, n + 1, 2))
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), (half, a, b)
    g.close()


def test_reference_free_lists_with_intervals_decoded_in_place(W, oracle):
    """Hand-assembled records again: every third node holds a list without reference made of residuals AND intervals (BVGraph.java:1042-1058)
    -- before, between and behind the residuals, adjacent ones, lists that are one interval only, lists long enough for skip entries --
    and the two nodes after it copy from it.  The scan kernel decodes the residuals of such a stored list straight into their places,
    shifted by the intervals they pass, and fills the intervals in where the residuals passed them (LongIntervalSequenceIterator.java:57-78)."""
    rng = np.random.default_rng(99)
    n = 6000
    recs, lists = [], []
    for x in range(n):
        lo, hi = max(0, x - 3000), min(n, x + 3000)
        if x % 3 == 0:
            kind = (x // 3) % 5
            nres = [0, 5, 40, 90, 20][kind]
            k = [1, 3, 2, 4, 1][kind]                                              # intervals
            pts = np.sort(rng.choice(np.arange(lo, hi - 40), size=k, replace=False))
            ivs, taken = [], set()
            for p in pts:
                ln = int(rng.integers(4, 30))
                if any(v in taken for v in range(int(p) - 1, int(p) + ln + 1)): continue   # keep intervals apart (adjacent ones would be one interval)
                ivs.append((int(p), ln)); taken.update(range(int(p), int(p) + ln))
            free = np.array([v for v in range(lo, hi) if v not in taken and (v - 1) not in taken and (v + 1) not in taken])
            res = sorted(int(v) for v in rng.choice(free, size=min(nres, len(free)), replace=False))
            # residuals must not form runs of 4 or more (the encoder would have made them an interval; a decoder does not care, but stay canonical)
            l = sorted(set(res) | taken)
            recs.append(Record(d=len(l), intervals=ivs, residuals=res))
        else:
            ref = x % 3
            base = lists[x - ref]
            k1 = len(base) // 3
            blocks, kept = ([k1], base[:k1]) if ref == 1 else ([0, k1], base[k1:])   # a prefix / everything but a prefix
            extra = sorted(int(v) for v in rng.choice(np.setdiff1d(np.arange(lo, hi), np.array(kept, dtype=np.int64)), size=int(rng.integers(0, 12)), replace=False))
            l = sorted(kept + extra)
            recs.append(Record(d=len(l), ref=ref, blocks=blocks, residuals=extra))
        lists.append(l)
    gbytes, offs, expect = assemble(recs)
    assert expect == lists
    p = W.default_params().clone(nodes=n, arcs=int(sum(len(l) for l in lists)))
    og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), gbytes, offs)
    g = W.BVGraph.from_memory(p, np.frombuffer(gbytes, dtype=np.uint8), offs)
    o = og.scan()
    for _ in range(3):
        r = g.scan()
        assert _same(r, o)
    assert r["lean_blocks"] >= 0.9 * (r["lean_blocks"] + r["slow_blocks"]), r
    for a, b in ((1, n - 1), (2000, 2007), (4001, 4003)):
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), (a, b)
    g.close()


def test_plan_cuts_blocks_around_large_lists(W, tools, oracle, monkeypatch):
    """The block plan puts a boundary in front of every list of 500 successors or more and 2 W + 1 nodes behind it, so that an LDS class
    (few wavefronts per CU) decodes only the nodes that touch such a list: more, smaller blocks -- and the same successors
    (BVGraph.java:995-1097 knows nothing of blocks: any cut must be invisible)."""
    n = 15000
    st = tools.synth_store(n, seed=61, synth=tools.eu_like(max_deg=30000, tail_alpha=1.6, mean_deg=40.0), threads=4)
    og = _og(oracle, st)
    o = og.scan()
    blocks = {}
    for cut in (True, False):
        if cut: monkeypatch.delenv("BVG_NO_LISTCUT", raising=False)
        else: monkeypatch.setenv("BVG_NO_LISTCUT", "1")
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
        for _ in range(3):
            r = g.scan()
            assert _same(r, o), cut
        blocks[cut] = r["lean_blocks"] + r["slow_blocks"]
        rng = np.random.default_rng(8)
        for _ in range(6):
            a, b = sorted(int(v) for v in rng.integers(0, n + 1, 2))
            ra, oa = g.scan(a, b), og.scan(a, b)
            assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), (cut, a, b)
        deg, succ = g.decode_range(0, n)
        odeg, osucc = og.decode_range(0, n)
        assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc), cut
        g.close()
    assert blocks[True] > blocks[False], blocks


# ---- the materialising form of the scan kernel (scan_kernel<..., MAT = true>): what backs nodeIterator() / successorBigArray()
# (BVGraph.java:1164-1176, SpeedTest.java:127-135) once the index exists ----------------------------------------------------------------
def _lean_blocks_of_last_decode(err):
    import re
    m = [l for l in err.splitlines() if "tiers concurrent" in l]
    assert m, err[-2000:]
    k = re.search(r"scan kernel (\d+) \+ (\d+)/(\d+)/(\d+)/(\d+) LDS-class", m[-1])
    return sum(int(v) for v in k.groups())


@pytest.mark.parametrize("shape", ["eu", "eu_dense", "web", "w0", "heavy_tail", "intervals", "no_d2"])
def test_materialising_scan_kernel_matches_oracle(W, tools, oracle, monkeypatch, capfd, shape):
    """Every list of every node, bit for bit, out of the lean kernel: lists that are copied from and lists without reference built in
    LDS and copied out, leaves with a reference merged straight into the output (MaskedLongIterator.java:73-100 + the extras, placed
    by position); whole graph, sub-ranges that cut blocks, a node base, a flyweight."""
    monkeypatch.setenv("BVG_DEBUG", "1"); monkeypatch.setenv("BVG_MAT_LEAN", "1")   # (sparse graphs materialise on the row kernel by default: here the lean kernel takes every shape)
    kw, synth, n = {}, None, 40000
    if shape == "eu": synth = tools.eu_like()
    if shape == "eu_dense": synth = tools.eu_like(mean_deg=127.5)
    if shape == "web": synth = tools.web_like(mean_deg=30.0)
    if shape == "w0": synth, kw = tools.web_like(mean_deg=40.0), dict(window_size=0, max_ref_count=0, min_interval_length=0)
    if shape == "heavy_tail": synth, n = tools.eu_like(max_deg=30000, tail_alpha=1.6, mean_deg=40.0), 15000
    if shape == "intervals": synth, kw = tools.eu_like(p_interval=0.9, interval_len=40.0), dict(min_interval_length=2)
    if shape == "no_d2": synth = tools.eu_like(p_interval=0.8); monkeypatch.setenv("BVG_NO_D2", "1")   # reference-free lists with intervals through the position tasks
    st = tools.synth_store(n, seed=47, params=W.default_params(**kw), synth=synth, threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = _og(oracle, st)
    odeg, osucc = og.decode_range(0, n)
    g.build_index()
    for rnd in range(2):                                               # (learned tiers settle after the first call)
        capfd.readouterr()
        deg, succ = g.decode_range(0, n)
        lean = _lean_blocks_of_last_decode(capfd.readouterr().err)
        assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc), (shape, rnd)
    assert lean > 0, "the materialising call did not run the lean kernel"
    cum = np.concatenate([[0], np.cumsum(odeg, dtype=np.int64)])
    rng = np.random.default_rng(9)
    for _ in range(10):
        a, b = sorted(int(v) for v in rng.integers(0, n + 1, 2))
        if b - a < 4096: b = min(n, a + 4096 + int(rng.integers(0, 3000)))   # (shorter ranges run index-less on the row kernel: other tests)
        d1, s1 = g.decode_range(a, b)
        assert np.array_equal(d1, odeg[a:b]) and np.array_equal(s1, osucc[cum[a]:cum[b]]), (shape, a, b)
    g.set_node_base((1 << 33) + 5)
    d1, s1 = g.decode_range(n // 4, n)
    assert np.array_equal(s1, osucc[cum[n // 4]:] + ((1 << 33) + 5))
    g.set_node_base(0)
    h = g.copy()
    d1, s1 = h.decode_range(0, n // 2)
    assert np.array_equal(s1, osucc[:cum[n // 2]])
    it = W.NodeIterator(h, n // 3, batch_nodes=8192)                    # the iterator's batches ride on it too
    for x in range(n // 3, n // 3 + 9000):
        assert it.next_long() == x
        assert np.array_equal(it.successor_array(), osucc[cum[x]:cum[x + 1]]), x
    it.close(); h.close(); g.close()


@pytest.mark.parametrize("pool", ["640", "1024", "3072"])
def test_materialising_scan_kernel_pool_sizes(W, tools, oracle, monkeypatch, pool):
    """Sub-rows cut short by the pool (the parked residuals of the leaves count in MAT mode), blocks failing over to the row kernel."""
    monkeypatch.setenv("BVG_SCAN_POOL", pool); monkeypatch.setenv("BVG_MAT_LEAN", "1")
    st = tools.synth_store(30000, seed=49, synth=tools.eu_like(mean_deg=100.0), threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    odeg, osucc = _og(oracle, st).decode_range(0, 30000)
    g.build_index()
    for _ in range(2):
        deg, succ = g.decode_range(0, 30000)
        assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc), pool
    g.close()


def test_materialising_scan_kernel_on_wide_graphs(W, tools, oracle, monkeypatch, capfd):
    """force_wide: 32-bit lists of ids relative to the block's base, the base added back on the way out."""
    monkeypatch.setenv("BVG_DEBUG", "1"); monkeypatch.setenv("BVG_MAT_LEAN", "1")
    st = tools.synth_store(30000, seed=51, synth=tools.eu_like(mean_deg=60.0), threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    g.set_tuning(force_wide=True)
    odeg, osucc = _og(oracle, st).decode_range(0, 30000)
    g.build_index()
    for _ in range(2):
        capfd.readouterr()
        deg, succ = g.decode_range(0, 30000)
        lean = _lean_blocks_of_last_decode(capfd.readouterr().err)
        assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc)
    assert lean > 0
    g.set_node_base((1 << 40) + 3)
    assert np.array_equal(g.decode_range(100, 29000)[1], osucc[np.cumsum(odeg, dtype=np.int64)[99]:np.cumsum(odeg, dtype=np.int64)[28999]] + ((1 << 40) + 3))
    g.close()


def test_cnr2000_materialised_by_the_lean_kernel_matches_the_golden(W, cnr_csr, monkeypatch, capfd):
    """BVGraphTest.testLarge through the materialising lean kernel (the reference's own expected lists)."""
    from conftest import CNR
    monkeypatch.setenv("BVG_DEBUG", "1"); monkeypatch.setenv("BVG_MAT_LEAN", "1")
    g = W.BVGraph.load(CNR)
    g.build_index()
    capfd.readouterr()
    deg, succ = g.decode_range(0, g.num_nodes())
    lean = _lean_blocks_of_last_decode(capfd.readouterr().err)
    gdeg, gsucc = cnr_csr
    assert np.array_equal(deg, gdeg) and np.array_equal(succ, gsucc) and lean > 0
    g.close()
