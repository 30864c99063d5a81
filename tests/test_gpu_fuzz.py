"""Randomised end-to-end parity: graph shape x BV parameters x forced tier (row kernels / giant kernel on every block / generic
kernel) x index on or off; every case compares scan and materialise against the CPU oracle and, for default codings, sends the
adjacency through the device compressor and back.  BVG_FUZZ=<n> runs n cases (default 16: a few seconds)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_graph(O, st):
    return O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)


def _adjacency(rng, n):
    """a random sparse graph with locality, a few long lists (some of them consecutive runs, some copying a neighbour)"""
    d = rng.poisson(float(rng.choice([1, 8, 40])), n).astype(np.int64)
    big = rng.choice(n, min(n, int(rng.integers(0, 4))), replace=False)
    lists = []
    for x in range(n):
        span = int(rng.choice([50, 2000, n]))
        lo, hi = max(0, x - span), min(n, x + span + 1)
        k = min(int(d[x]), hi - lo)
        if x in big:
            k = min(n, int(rng.choice([3000, 15000, 40000])))
            l = np.sort(rng.choice(n, k, replace=False)) if rng.random() < 0.7 else np.arange(int(rng.integers(0, max(1, n - k))), 0)[:0]
            if l.size == 0:
                s0 = int(rng.integers(0, max(1, n - k))); l = np.arange(s0, s0 + k)
        elif lists and rng.random() < 0.3 and lists[-1].size:
            prev = lists[-1]; keep = prev[rng.random(prev.size) < 0.7]
            l = np.union1d(keep, rng.integers(lo, hi, max(0, k // 3)))
        else:
            l = np.unique(rng.integers(lo, hi, k)) if k else np.empty(0, np.int64)
        lists.append(l.astype(np.int64))
    off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum([l.size for l in lists])
    return off, (np.concatenate(lists) if n and off[-1] else np.empty(0, np.int64))


def test_random_graphs_parameters_and_tiers(W, tools, oracle, monkeypatch):
    cases = int(os.environ.get("BVG_FUZZ", "16"))
    rng = np.random.default_rng(int(os.environ.get("BVG_FUZZ_SEED", "7")))
    first = int(os.environ.get("BVG_FUZZ_FROM", "0"))
    for case in range(cases):
        for k in ("BVG_GIANT", "BVG_NOSKIP", "BVG_EMIT", "BVG_DBG"):
            monkeypatch.delenv(k, raising=False)
        n = int(rng.choice([1, 70, 900, 6000, 45000]))
        kw = dict(window_size=int(rng.choice([0, 1, 3, 7, 20, 70])), max_ref_count=int(rng.choice([0, 1, 3, 50, -1])),
                  min_interval_length=int(rng.choice([0, 2, 4, 7])), zeta_k=int(rng.choice([1, 2, 3, 5])))
        if rng.random() < 0.3:                                                 # non-default codings: the generic field decoders
            kw.update(outdegree_coding=int(rng.choice([1, 2])), block_coding=int(rng.choice([1, 2, 5])), residual_coding=int(rng.choice([1, 2, 3, 6, 7])),
                      reference_coding=int(rng.choice([1, 2, 5])), block_count_coding=int(rng.choice([1, 2, 5])))
            if kw["residual_coding"] == 3: kw["zeta_k"] = int(rng.choice([1, 3, 5, 8]))
        tier = str(rng.choice(["default", "giant", "giant", "tasks", "pipelined", "generic"]))
        env = {"giant": dict(BVG_GIANT="2"), "tasks": dict(BVG_EMIT="1", BVG_DBG="16"), "pipelined": dict(BVG_EMIT="0")}.get(tier, {})
        if rng.random() < 0.3: env["BVG_NOSKIP"] = "1"
        for k, v in env.items(): monkeypatch.setenv(k, v)
        off, adj = _adjacency(rng, n)
        p = W.default_params(**kw)
        if case < first:                                                       # BVG_FUZZ_FROM=<case>: replay the generator up to a case (every draw below is made, nothing is run)
            rng.random(); rng.integers(0, n + 1, 2); rng.integers(0, n, min(n, 40))
            if n <= 6000 and adj.size: rng.random()
            rng.choice([0, 64, 1000])
            continue
        st = tools.store((off, adj), p, threads=2)
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
        wide = bool(rng.random() < 0.2)                                        # the 64-bit successor kernels on a small graph
        if tier == "generic" or wide: g.set_tuning(force_slow=tier == "generic", force_wide=wide)
        og = _oracle_graph(oracle, st)
        what = (case, n, kw, tier, env, wide)
        o = og.scan()
        for _ in range(2):                                                     # the second scan uses the index the first one built
            try:
                r = g.scan()
            except Exception:
                print("fuzz case that raised:", what, flush=True)
                raise
            assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"]), what
        deg, succ = g.decode_range(0, n)
        assert np.array_equal(deg, np.diff(off.astype(np.int64))) and np.array_equal(succ, adj), what
        a, b = sorted(int(v) for v in rng.integers(0, n + 1, 2))
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"]), what + (a, b)
        nodes = rng.integers(0, n, min(n, 40)).astype(np.int64)
        bd, bs = g.successors_batch(nodes)
        assert np.array_equal(bs, np.concatenate([adj[int(off[x]):int(off[x + 1])] for x in nodes]) if len(nodes) else bs), what
        if n <= 6000 and adj.size and rng.random() < 0.5:                      # the transposition feed (Transform.transposeOffline on the device)
            src = np.repeat(np.arange(n, dtype=np.int64), np.diff(off.astype(np.int64)))
            order = np.lexsort((src, adj))
            toff, tsucc = g.transpose()
            assert np.array_equal(tsucc, src[order]), what
            assert np.array_equal(toff, np.concatenate([[0], np.cumsum(np.bincount(adj, minlength=n))]).astype(np.uint64)), what
        g.close()
        # the device compressor writes the very bytes the CPU tooling wrote
        chunk = int(rng.choice([0, 64, 1000]))
        if kw["window_size"] <= 64:                                            # (the device compressor's window limit is 127; the tooling's chunked form matches BVG:2404-2457)
            ref = st if chunk == 0 else tools.store((off, adj), p, chunk_nodes=chunk, threads=2)
            gb, go = W.store((off, adj), p, chunk_nodes=chunk)
            assert np.array_equal(gb, ref.graph) and np.array_equal(go, ref.offsets), what + (chunk,)
        if case % 100 == 99: print("fuzz: %d of %d cases" % (case + 1, cases), flush=True)      # (long runs: `pytest -s` shows progress)
