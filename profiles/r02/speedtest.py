#!/usr/bin/env python3
"""The reference's SpeedTest protocol (test/SpeedTest.java:87-141) on the drop-in API, i.e. what a host caller gets END TO END
(successors delivered in HOST memory, PCIe included), next to the CPU port on this box's cores:
  sequential: nodeIterator() over every node, outdegree() + successorBigArray() per node (SpeedTest.java:127-141) through the
              pipelined NodeIterator (bvg_decode_range batches into page-locked buffers, next batch decoded while this one is read);
  random:     successors(x) of `samples` random nodes (SpeedTest.java:87-116) through bvg_successors_batch in frontiers.
WARMUP 1 + REPEAT 3 here (the reference: 3 + 10).  usage: speedtest.py [--shape eu] [--gib 1] [--samples 2000000]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="eu"); ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--samples", type=int, default=2000000); ap.add_argument("--frontier", type=int, default=1 << 18)
    ap.add_argument("--batch-nodes", type=int, default=1 << 20)
    args = ap.parse_args()
    import webgraph_big_amd as W
    import tooling as T
    from oracle import bvg_oracle as O
    import bench as B
    kind, skw, pkw, _, wl = B.SHAPES[args.shape]
    synth = T.eu_like(**skw) if kind == "eu" else T.web_like(**skw)
    threads = min(os.cpu_count() or 1, 64)
    st = T.synth_store(1 << 21, seed=0, params=W.default_params(**pkw), synth=synth, threads=threads)
    copies = max(1, int(args.gib * (1 << 30) / len(st.graph)))
    base = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    g = base.tile(copies) if copies > 1 else base
    n, m = g.num_nodes(), st.stats["arcs"] * copies
    og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    out = {"workload": wl, "nodes": n, "arcs": m, "graph_bytes": len(st.graph) * copies}
    g.scan()                                                   # plan + skip index (load-time work, not timed)

    # ---- sequential, host path ----
    times = []
    for k in range(4):
        t0 = time.perf_counter()
        it = W.NodeIterator(g, 0, batch_nodes=args.batch_nodes)
        tot = 0; z = 0
        while it.has_next():
            it.next_long()
            b0, deg, cum, succ = it.batch()                   # every node of the batch: outdegree() and successorBigArray() are views into these
            tot += int(deg.sum(dtype=np.int64)); z ^= int(succ[-1]) if len(succ) else 0
            it.skip_batch()
        it.close()
        dt = time.perf_counter() - t0
        assert tot == m, (tot, m)
        if k >= 1:
            times.append(dt)
    t = float(np.mean(times))
    # parity of what arrived: first tile against the oracle
    it = W.NodeIterator(g, 0, upper_bound=st.params.nodes, batch_nodes=1 << 16)
    odeg, osucc = og.decode_range(0, 1 << 16)
    it.next_long(); b0, deg, cum, succ = it.batch()
    assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc)
    it.close()
    out["sequential_host"] = {"edges_per_s": m / t, "nodes_per_s": n / t, "s_per_pass": t, "host_GB_per_s": (succ.dtype.itemsize * m + 4.0 * n) / t / 1e9,
                              "batch_nodes": args.batch_nodes, "note": "NodeIterator batches, successors as uint32 (ids below 2^32; int64 otherwise) in page-locked host memory; decode of batch i+1 overlaps the walk over batch i"}
    # the same without the pipeline and with pageable buffers (the round-1 path)
    t0 = time.perf_counter(); tot = 0
    for lo in range(0, n, args.batch_nodes):
        deg, succ = g.decode_range(lo, min(lo + args.batch_nodes, n)); tot += len(succ)
    t1 = time.perf_counter() - t0
    out["sequential_host_unpipelined_pageable"] = {"edges_per_s": tot / t1, "s_per_pass": t1}

    # ---- random access ----
    rng = np.random.default_rng(0x5eed)
    nodes = rng.integers(0, n, size=args.samples, dtype=np.int64)
    times = []; tot = 0
    for k in range(4):
        t0 = time.perf_counter(); tot = 0
        for i in range(0, len(nodes), args.frontier):
            deg, succ = g.successors_batch(nodes[i:i + args.frontier]); tot += len(succ)
        dt = time.perf_counter() - t0
        if k >= 1:
            times.append(dt)
    t = float(np.mean(times))
    # parity on a sample + the CPU port's single-thread rate (the reference's SpeedTest is single-threaded)
    n0 = st.params.nodes
    sub = nodes[:20000]
    deg, succ = g.successors_batch(sub)
    t0 = time.perf_counter(); cpu_arcs = 0
    for x in sub:
        cpu_arcs += len(og.successors(int(x % n0)))
    tc = time.perf_counter() - t0
    cum = np.concatenate([[0], np.cumsum(deg)])
    for j in range(0, 2000):
        x = int(sub[j]); s = og.successors(x % n0) + (x // n0) * n0
        assert np.array_equal(s, succ[cum[j]:cum[j + 1]]), j
    out["random_access"] = {"samples": args.samples, "frontier": args.frontier, "nodes_per_s": args.samples / t, "arcs_per_s": tot / t, "s_per_pass": t,
                            "cpu_port_1thread": {"nodes_per_s": len(sub) / tc, "arcs_per_s": cpu_arcs / tc, "samples": len(sub)},
                            "note": "bvg_successors_batch, successors in host memory; each request decodes its reference chain (halo) too"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
