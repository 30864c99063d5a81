"""The scan's checksum pays for every PRODUCED successor (include/bvgraph_hip.h, "WHAT THE SCAN MUST DO"): one multiply-add per decoded residual, per
interval element and per element a copy block keeps -- never a closed form over a run.

Hand-assembled records (tests/bvrecords.py): groups of one base list (a stored list: later nodes copy from it) and nodes that copy from it through every
kind of mask (MaskedLongIterator.java:73-100: all of it, odd / even block counts, an empty first block), as leaves and as stored lists of a chain
(BVG:1062-1090), with and without intervals and residuals of their own.  Stream B differs from stream A in ONE element of ONE base list (+1).  On both
streams the HIP scan agrees with the oracle; and, node by node through the lean scan kernel (validated blocks, the second scan onwards), every node that
holds the changed element -- by copying it, directly or down the chain -- moves by exactly k1(node) * 1, every other node not at all.  The test cannot tell a
closed form from a sum (they are equal by definition); it pins that no copy is skipped, cached across streams or attributed to the wrong node."""
import numpy as np
import pytest

from bvrecords import Record, assemble

M64 = (1 << 64) - 1


def _k1(x):
    m = 0xFFFFFFFF
    h = ((x & m) * 0x9E3779B1 + (x >> 32) * 0x85EBCA77) & m
    h ^= h >> 15; h = (h * 0x2C1B3C6D) & m; h ^= h >> 12
    return h | 1


GROUP = 12          # nodes per group
NRES = 40           # residuals of a base list: >= 16, so the skip index cuts it into tasks (the lean kernel's path)


def _group(b, bump=None):
    """Records of the group whose base list is node b.  bump = index of the base element that is one larger (stream B)."""
    base = [b + 5 + 7 * i for i in range(NRES)]                      # gaps of 7: +1 keeps the list sorted and duplicate-free
    if bump is not None:
        base[bump] += 1
    L = {0: base}
    recs = [Record(d=NRES, residuals=base)]
    # 1: copies all of it (no blocks: MaskedLongIterator.java:73-78) -- a leaf
    recs.append(Record(d=NRES, ref=1, blocks=[]))
    L[1] = list(base)
    # 2: odd block count (keep 10, skip 5, keep 10; tail dropped) + two residuals of its own -- a STORED list (node 3 copies from it)
    l2 = sorted(base[0:10] + base[15:25] + [b + 2, b + 1000])
    recs.append(Record(d=len(l2), ref=2, blocks=[10, 5, 10], residuals=[b + 2, b + 1000]))
    L[2] = l2
    # 3: copies node 2 (chain depth 2) with an even block count (keep 4, skip 3, keep the rest) and an interval of 5 -- a leaf with a reference to a stored list with reference
    kept3 = l2[0:4] + l2[7:]
    iv = (b + 2000, 5)
    l3 = sorted(kept3 + [iv[0] + k for k in range(iv[1])])
    recs.append(Record(d=len(l3), ref=1, blocks=[4, 3], intervals=[iv]))
    L[3] = l3
    # 4: empty first block (skip 3, keep 20, skip the tail: blocks [0, 3, 20]) of the base -- a leaf
    l4 = base[3:23]
    recs.append(Record(d=len(l4), ref=4, blocks=[0, 3, 20]))
    L[4] = l4
    # 5: every other element in blocks of one (20 blocks), plus 17 residuals (a task list of its own) -- stored (node 6 copies it)
    own5 = [b + 3000 + 3 * k for k in range(17)]
    kept5 = [base[2 * k] for k in range(10)] + base[20:]
    l5 = sorted(kept5 + own5)
    recs.append(Record(d=len(l5), ref=5, blocks=[1] * 20, residuals=own5))
    L[5] = l5
    # 6: all of node 5 -- a leaf (chain depth 2 through the one-element blocks)
    recs.append(Record(d=len(l5), ref=1, blocks=[]))
    L[6] = list(l5)
    # 7 .. GROUP-1: plain fillers
    for j in range(7, GROUP):
        recs.append(Record(d=6, residuals=[b + j + 3 + 5 * i for i in range(6)]))
        L[j] = list(recs[-1].residuals)
    return recs, L


def _stream(ngroups, bump_group=None, bump=None):
    recs, lists = [], []
    for g in range(ngroups):
        r, L = _group(g * GROUP, bump if g == bump_group else None)
        recs += r
        lists += [L[j] for j in range(GROUP)]
    graph, offs, exp = assemble(recs)
    assert [list(l) for l in exp] == lists, "the hand-computed lists disagree with the restated iterators"
    return recs, np.frombuffer(graph, dtype=np.uint8), offs, lists


@pytest.mark.gpu
@pytest.mark.parametrize("bump", [0, 7, 17, 22, 39])
def test_every_copy_of_a_changed_element_moves_the_checksum(W, oracle, bump):
    ngroups = 600                                                    # 7 200 nodes: the first scan builds the skip index and validates the blocks
    bg = 311
    _, ga, offa, la = _stream(ngroups)
    _, gb, offb, lb = _stream(ngroups, bg, bump)
    n = ngroups * GROUP
    p = W.default_params().clone(nodes=n, arcs=int(sum(len(l) for l in la)))
    res = []
    for g, offs in ((ga, offa), (gb, offb)):
        og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), g.tobytes(), offs)
        hg = W.BVGraph.from_memory(p, g, offs)
        o = og.scan()
        r1 = hg.scan(); r2 = hg.scan()                               # the second scan runs the lean kernel on the validated blocks
        assert (r1["nodes"], r1["arcs"], r1["chk"]) == (o["nodes"], o["arcs"], o["chk"])
        assert (r2["nodes"], r2["arcs"], r2["chk"]) == (o["nodes"], o["arcs"], o["chk"])
        assert r2["lean_blocks"] > 0, "the lean scan kernel did not run"
        res.append((og, hg, o))
    (oga, hga, oa), (ogb, hgb, ob) = res
    v = la[bg * GROUP][bump]
    holders = 0
    for j in range(GROUP):
        y = bg * GROUP + j
        ra, rb = hga.scan(y, y + 1), hgb.scan(y, y + 1)
        assert ra["lean_blocks"] > 0 and rb["lean_blocks"] > 0
        assert (ra["arcs"], rb["arcs"]) == (len(la[y]), len(lb[y]))
        assert ra["chk"] == oga.scan(y, y + 1)["chk"] and rb["chk"] == ogb.scan(y, y + 1)["chk"]
        holds = v in la[y]
        assert holds == (la[y] != lb[y])
        if holds:
            holders += 1
            assert (rb["chk"] - ra["chk"]) & M64 == _k1(y), "node %d copies the changed element: its checksum must move by k1 * 1" % y
        else:
            assert rb["chk"] == ra["chk"]
    assert holders >= 2, "the changed element should be held by the base list and at least one copier"
    # the whole-graph sums differ by the sum of the holders' keys
    want = sum(_k1(bg * GROUP + j) for j in range(GROUP) if v in la[bg * GROUP + j]) & M64
    assert (ob["chk"] - oa["chk"]) & M64 == want
    for _, hg, _ in res:
        hg.close()


def test_integrity_streams_decode_to_the_hand_computed_lists(W, oracle):
    """CPU half: the oracle decodes both hand-assembled streams to the lists computed by hand, and the two differ exactly where the changed element is held."""
    _, ga, offa, la = _stream(40)
    _, gb, offb, lb = _stream(40, 13, 17)
    p = W.default_params().clone(nodes=40 * GROUP, arcs=int(sum(len(l) for l in la)))
    for g, offs, L in ((ga, offa, la), (gb, offb, lb)):
        og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), g.tobytes(), offs)
        deg, succ = og.decode_range(0, p.nodes)
        assert deg.tolist() == [len(l) for l in L] and succ.tolist() == [x for l in L for x in l]
    v = la[13 * GROUP][17]
    diff = [y for y in range(40 * GROUP) if la[y] != lb[y]]
    assert diff == [13 * GROUP + j for j in range(GROUP) if v in la[13 * GROUP + j]] and len(diff) >= 2
