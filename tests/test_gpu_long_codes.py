"""Residual codes of 32 bits and more (zeta_3 gaps of 2^21 and above: BVG:788-795) through the lean scan kernel's step loop (csrc/bvg_scan_steps3.inc): its fast
path decodes from a 32-bit window and hands such a code to the 64-bit decoder -- as the first or the second code of a pair, as the odd last code, as a list's first
(signed, BVG:917) gap, in lists cut into tasks by the skip index and in lists short enough for the one-task-per-lane loop, in stored lists and in leaves.
Hand-assembled records (tests/bvrecords.py); the oracle is the reference."""
import numpy as np
import pytest

from bvrecords import Record, assemble

BIG = 1 << 22          # a gap of 2^22: a zeta_3 code of 32 bits (h = 7)
HUGE = 1 << 29         # 2^29: 40 bits


def _lists(x):
    """A few lists for node x, every kind of position for a long code."""
    L = []
    # 40 residuals, long codes at even and odd indices (the skip index cuts at every 16th)
    v, l = x + 3, []
    for i in range(40):
        l.append(v); v += (BIG + 5 * i) if i in (3, 8, 15, 16, 30, 31, 35) else (HUGE if i in (21, 37) else 2 + (i % 5))
    L.append(l)
    # first gap long and positive / negative (nat2int of a large value), then short gaps
    L.append([x + BIG * 3 + k * 3 for k in range(20)])
    L.append([7 + k * 2 for k in range(18)] if x > BIG else [x + HUGE + k for k in range(18)])
    # short lists (the one-task-per-lane loop): 1, 2, 3 residuals, the long code first / last / alone
    L.append([x + HUGE])
    L.append([x + 1, x + 1 + BIG])
    L.append([x + 2, x + 2 + BIG, x + 9 + BIG])
    L.append([x + BIG, x + BIG + HUGE, x + BIG + HUGE + 1])
    return L


def _graph(groups):
    recs, x = [], 0
    for g in range(groups):
        Ls = _lists(x)
        base = Ls[0]
        recs.append(Record(d=len(base), residuals=base))
        # a leaf copying the first list of the group through a mask, with a long residual of its own
        recs.append(Record(d=len(base) - 4 + 1, ref=1, blocks=[10, 4], residuals=[base[-1] + BIG]))
        # a stored copy of the same list with extras (one of them behind a 40-bit code), copied again
        recs.append(Record(d=12, ref=2, blocks=[10], residuals=[base[9] + 1, base[-1] + HUGE]))
        recs.append(Record(d=12, ref=1, blocks=[]))
        x += 4
        for l in _lists(x)[1:]:
            recs.append(Record(d=len(l), residuals=l)); x += 1
    return recs


def _run(W, oracle, groups):
    recs = _graph(groups)
    g, offs, lists = assemble(recs)
    n = len(recs)
    p = W.default_params().clone(nodes=n, arcs=int(sum(r.d for r in recs)))
    gb = np.frombuffer(g, dtype=np.uint8)
    og = oracle.Graph.from_memory(oracle.Params(**p.as_dict()), g, offs)
    deg, succ = og.decode_range(0, n)
    assert succ.tolist() == [v for l in lists for v in l]
    return p, gb, offs, og, lists


def test_oracle_decodes_the_long_codes(W, oracle):
    _run(W, oracle, 30)


@pytest.mark.gpu
@pytest.mark.parametrize("no_index", [0, 2])
def test_lean_kernel_takes_codes_of_32_bits_and_more(W, oracle, no_index):
    """no_index = 2 (marks only): the lists of 40 residuals have no skip entries and are decoded by the whole wavefront behind their first 32 residuals (bvg_scan.hip, "COOP"):
    the long codes at residuals 35 and 37 go through its 64-bit detour."""
    p, gb, offs, og, lists = _run(W, oracle, 500)                      # 5 000 nodes: the first scan builds the index, the later ones run the lean kernel
    hg = W.BVGraph.from_memory(p, gb, offs)
    if no_index: hg.set_tuning(no_index=no_index)
    o = og.scan()
    for i in range(3):
        r = hg.scan()
        assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"]), i
    assert r["lean_blocks"] > 0
    deg, succ = hg.decode_range(0, p.nodes)
    assert succ.tolist() == [v for l in lists for v in l]
    a, b = 1003, 2511                                                  # a node range: blocks cut by the range, halos
    r2, o2 = hg.scan(a, b), og.scan(a, b)
    assert (r2["arcs"], r2["chk"]) == (o2["arcs"], o2["chk"])
    hg.close()
