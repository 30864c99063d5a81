"""Streams the encoder never writes: the dedup-on-equal-heads and cap-at-d branches of
MergedLongIterator (MergedLongIterator.java:64-65,85-89), copy blocks that over-run the referenced list
(MaskedLongIterator.java:93-100), lists that come out short and are padded with -1 (BVGraph.java:1171), and a reference to
such a short list.  Records are hand-assembled bit by bit (tests/bvrecords.py); the expected answer is a literal Python
restatement of the reference's iterators, which pins the C oracle on these cases (CPU test), and the HIP path must agree with
the oracle in every emission mode (GPU test), taking its literal-iterator tier where the position tasks cannot apply.

Mirrors the reference's own unit tests of the iterators (test/.../MergedLongIteratorTest.java:28-63,
MaskedLongIteratorTest.java:29-106) at the level of whole records.
"""
import numpy as np
import pytest

from bvrecords import Record, assemble


def _filler(x, k=6):
    """A plain well-formed node: k residuals after x."""
    return Record(d=k, residuals=[x + 3 + 5 * i for i in range(k)])


def _cases():
    """name -> list of Records; node 0 is always a plain 10-element list 10,20,...,100 (the referenced list)."""
    base = Record(d=10, residuals=[10 * (i + 1) for i in range(10)])
    C = {}
    # residual equal to a copied element: emitted once, list one short (-1 at the end)
    C["residual_equals_copied"] = [base, Record(d=6, ref=1, blocks=[4], residuals=[20, 35])]
    # interval overlapping a residual (inner merge dedups): intervals [50..53], residual 52
    C["interval_overlaps_residual"] = [base, Record(d=9, ref=1, blocks=[3], intervals=[(50, 4)], residuals=[52, 77])]
    # interval overlapping a copied element: copied 10..40, interval [38..41] holds 40
    C["interval_overlaps_copied"] = [base, Record(d=9, ref=1, blocks=[4], intervals=[(38, 4)], residuals=[99])]
    # blocks over-running the referenced list, odd count: keep 3, skip 2, keep 9 (only 5 left)
    C["blocks_overrun_odd"] = [base, Record(d=13, ref=1, blocks=[3, 2, 9], residuals=[7])]
    # blocks over-running, even count: copied = 3 + 4 + (10 - 15) = 2 by BVG:1030, but the mask yields 3 + 4 = 7 elements:
    # with d = 6 the outer merge stops at d with elements left (cap at d, MergedLongIterator.java:64-65)
    C["cap_at_d"] = [base, Record(d=6, ref=1, blocks=[3, 2, 4, 6], residuals=[5, 15, 25, 1000])]
    # everything at once + equal heads between all three streams
    C["all_three_equal"] = [base, Record(d=12, ref=1, blocks=[5], intervals=[(28, 5)], residuals=[30, 31])]
    # a reference to a list that came out short (its tail holds -1): node 1 is short by one, node 2 copies all of it
    C["reference_to_short_list"] = [base, Record(d=6, ref=1, blocks=[4], residuals=[20, 35]),
                                    Record(d=8, ref=1, blocks=[], residuals=[1, 2])]
    # chain of depth 3 over a deduplicated list
    C["chain_over_dedup"] = [base, Record(d=6, ref=1, blocks=[4], residuals=[20, 35]), Record(d=6, ref=1, blocks=[2, 1], residuals=[36]),
                             Record(d=4, ref=1, blocks=[1, 1, 1], residuals=[35, 37])]
    # first block empty, over-run inside a skip block
    C["skip_overrun"] = [base, Record(d=4, ref=1, blocks=[0, 4, 2, 30, 1], residuals=[1])]
    # duplicate residual next to an interval end, no reference (inner merge only)
    C["pure_interval_residual_dup"] = [base, Record(d=7, intervals=[(200, 4)], residuals=[203, 204, 300])]
    return C


CASES = _cases()


def _graph(case, pad_nodes=0, lead_nodes=0):
    """Optionally surrounds the case with plain nodes so that it sits in the middle of a row / at a block edge."""
    recs = list(CASES[case])
    shift = lead_nodes
    if shift:
        # leading nodes first; the case's values are absolute, references relative, so only ids move
        recs = [_filler(i) for i in range(shift)] + recs
    recs = recs + [_filler(len(recs) + i) for i in range(pad_nodes)]
    return recs


def _store(W, recs):
    g, offs, lists = assemble(recs)
    n = len(recs)
    p = W.default_params().clone(nodes=n, arcs=int(sum(r.d for r in recs)))
    return p, np.frombuffer(g, dtype=np.uint8), offs, lists


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("lead,pad", [(0, 0), (5, 80), (70, 3)])
def test_oracle_follows_the_reference_iterators(W, oracle, case, lead, pad):
    p, g, offs, lists = _store(W, _graph(case, pad, lead))
    og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), g.tobytes(), offs)
    deg, succ = og.decode_range(0, p.nodes)
    assert deg.tolist() == [len(l) for l in lists]
    assert succ.tolist() == [v for l in lists for v in l]
    for x in range(p.nodes):                                  # random access: recursive chains (BVG:1084)
        assert og.successors(x).tolist() == lists[x]
    if case != "pure_interval_residual_dup":
        assert any(-1 in l for l in lists) or case == "cap_at_d", "the case should leave a short list"


MODES = {
    "default": {},
    "tasks_every_row": dict(BVG_EMIT="1", BVG_DBG="16"),
    "pipelined_rows_in_task_variant": dict(BVG_EMIT="1", BVG_DBG="32"),
    "pipelined_variant": dict(BVG_EMIT="0"),
    "giant_kernel_every_block": dict(BVG_GIANT="2"),          # (csrc/bvg_giant.hip: it must refuse these records, the generic kernel takes them)
}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("case", sorted(CASES))
def test_hip_matches_oracle_on_streams_the_encoder_never_writes(W, oracle, monkeypatch, case, mode):
    for k in ("BVG_EMIT", "BVG_DBG", "BVG_NOSKIP", "BVG_GIANT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    for lead, pad in [(0, 0), (5, 80), (70, 3), (0, 300)]:
        p, g, offs, lists = _store(W, _graph(case, pad, lead))
        og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), g.tobytes(), offs)
        hg = W.BVGraph.from_memory(p, g, offs)
        o, r = og.scan(), hg.scan()
        assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"]), (case, mode, lead, pad)
        deg, succ = hg.decode_range(0, p.nodes)
        assert deg.tolist() == [len(l) for l in lists]
        assert succ.tolist() == [v for l in lists for v in l], (case, mode, lead, pad)
        x = lead + len(CASES[case]) - 1                       # the odd node alone (random access: its chain is the halo)
        d1, s1 = hg.decode_range(x, x + 1)
        assert s1.tolist() == lists[x]
        sb = hg.successors_batch(np.array([x, lead], dtype=np.int64))
        assert sb[1].tolist() == lists[x] + lists[lead]
        if mode == "tasks_every_row" and case not in ("pure_interval_residual_dup",):
            # the position tasks assume disjoint streams: these rows must have been handed to the literal-iterator tier
            assert r["slow_blocks"] > 0, "the fallback tier did not run (%s)" % case
        hg.close()


@pytest.mark.gpu
def test_many_odd_nodes_among_ordinary_ones(W, oracle, tools):
    """Odd records sprinkled over a real synthetic graph: every block that holds one falls back, the others do not."""
    st = tools.synth_store(3000, seed=5, synth=tools.web_like(), threads=2)
    og0 = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    deg0, succ0 = og0.decode_range(0, 3000)
    # rebuild the graph record by record from its decoded lists (no compression: residuals only), with odd nodes mixed in
    cum = np.concatenate([[0], np.cumsum(deg0)])
    recs = []
    for x in range(3000):
        l = succ0[cum[x]:cum[x + 1]].tolist()
        if x % 97 == 50 and len(recs) and recs[-1].d >= 4:    # copies the first 3 of the previous list and repeats one of them
            prev = lists_prev
            recs.append(Record(d=5, ref=1, blocks=[3], residuals=[prev[1], prev[2] + 1 if prev[2] + 1 not in prev else prev[-1] + 7]))
            lists_prev = None
        else:
            recs.append(Record(d=len(l), residuals=l))
            lists_prev = l
    g, offs, lists = assemble(recs)
    p = W.default_params().clone(nodes=3000, arcs=int(sum(r.d for r in recs)))
    gb = np.frombuffer(g, dtype=np.uint8)
    og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), g, offs)
    hg = W.BVGraph.from_memory(p, gb, offs)
    o, r = og.scan(), hg.scan()
    assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])
    deg, succ = hg.decode_range(0, 3000)
    assert succ.tolist() == [v for l in lists for v in l]
    hg.close()


# ---- the oracle's ONE documented deviation from the Java (oracle/bvg_oracle.c, "we treat <= 0 as none") --------------------------
# A record whose copied elements alone exceed its outdegree has a NEGATIVE residual count (BVG:1036,1062: extra = d - copied).  The
# Java builds a ResidualLongIterator whose `remaining` never reaches 0 (BVG:902-935): it would read residual codes out of whatever bits
# follow the record until the merge has produced d values -- a result that depends on the NEXT records' bits.  The oracle (and with it
# the HIP path) treats a count <= 0 as "no residuals": the list is the first d kept elements of the referenced list.  Pinned here on
# both sides so the deviation cannot drift.
def _negative_count_graph(W, lead=0, pad=0):
    base = Record(d=10, residuals=[10 * (i + 1) + lead for i in range(10)])
    odd = Record(d=3, ref=1, blocks=[])                       # no blocks: the whole referenced list is kept -> copied = 10 > d = 3, extra = -7
    odd2 = Record(d=2, ref=1, blocks=[1, 2])                  # even block count: keep 1, skip 2, keep the other 7 -> copied = 8 > d = 2, extra = -6
    recs = [_filler(i) for i in range(lead)] + [base, odd, base_like(lead + 2), odd2] + [_filler(lead + 4 + i) for i in range(pad)]
    w_lists = []
    from bvrecords import PyBits
    w = PyBits(); offs = [0]
    for x, rec in enumerate(recs):
        extra = rec.d
        if rec.d and rec.ref > 0:
            total = sum(rec.blocks); copied = sum(rec.blocks[0::2])
            if len(rec.blocks) % 2 == 0:
                copied += len(w_lists[x - rec.ref]) - total
            extra = rec.d - copied
        rec.write(w, x, 7, 4, 3, extra)
        offs.append(len(w))
        if rec.ref > 0 and extra < 0:                         # the documented deviation: no residuals, the first d kept elements
            ref = w_lists[x - rec.ref]; keep = []; pos = 0
            for i, b in enumerate(rec.blocks):
                if i % 2 == 0:
                    keep += ref[pos:pos + b]
                pos += b
            if len(rec.blocks) % 2 == 0:
                keep += ref[pos:]
            w_lists.append(keep[:rec.d])
        else:
            w_lists.append(list(rec.residuals))
    p = W.default_params().clone(nodes=len(recs), arcs=int(sum(r.d for r in recs)))
    return p, np.frombuffer(w.tobytes(), dtype=np.uint8), np.array(offs, dtype=np.uint64), w_lists


def base_like(x):
    return Record(d=10, residuals=[1000 + x + 7 * i for i in range(10)])


@pytest.mark.parametrize("lead,pad", [(0, 0), (5, 80), (70, 3)])
def test_oracle_treats_a_negative_residual_count_as_none(W, oracle, lead, pad):
    p, g, offs, lists = _negative_count_graph(W, lead, pad)
    og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), g.tobytes(), offs)
    deg, succ = og.decode_range(0, p.nodes)
    assert deg.tolist() == [len(l) for l in lists]
    assert succ.tolist() == [v for l in lists for v in l]
    assert lists[lead + 1] == [10 + lead, 20 + lead, 30 + lead] and len(lists[lead + 3]) == 2
    for x in range(p.nodes):
        assert og.successors(x).tolist() == lists[x]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", sorted(MODES))
def test_hip_refuses_a_negative_residual_count(W, oracle, monkeypatch, mode):
    """The other side of the deviation: the HIP path does NOT guess.  A record whose counts contradict each other is a malformed file
    (bounded, not bit-exact): every call whose node range holds such a record ends with EOFException (BVG_E_EOF, the
    ERR_MALFORMED bit) in every tier -- never a hang (the Java's iterator would not stop), never a silently different list -- while
    ranges whose node blocks do not hold it decode exactly."""
    for k in ("BVG_EMIT", "BVG_DBG", "BVG_NOSKIP", "BVG_GIANT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    for lead, pad in [(0, 0), (5, 80), (70, 3), (0, 5000)]:
        p, g, offs, lists = _negative_count_graph(W, lead, pad)
        hg = W.BVGraph.from_memory(p, g, offs)
        for _ in range(2):                                    # (twice: the second scan of the large case would run indexed)
            with pytest.raises(W.EOFException):
                hg.scan()
        with pytest.raises(W.EOFException):
            hg.decode_range(0, p.nodes)
        for x in (lead + 1, lead + 3):
            with pytest.raises(W.EOFException):
                hg.decode_range(x, x + 1)
            with pytest.raises(W.EOFException):
                hg.successors_batch(np.array([x], dtype=np.int64))
        # errors are reported per node BLOCK (the unit a wavefront decodes): a range whose blocks do not hold the odd records is exact
        if pad > 3000:
            flat = lambda a, b: [v for l in lists[a:b] for v in l]
            a0 = 2000
            assert hg.decode_range(a0, p.nodes)[1].tolist() == flat(a0, p.nodes)
            og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), g.tobytes(), offs)
            r, o = hg.scan(a0, p.nodes), og.scan(a0, p.nodes)
            assert (r["arcs"], r["chk"]) == (o["arcs"], o["chk"])
        hg.close()
