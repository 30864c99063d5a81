# r02: how the scan copes with giant lists (transposed / social graphs: lists of 10^5..10^7 successors).  A graph of N ordinary
# nodes (~10 successors) with G giants of D successors each (half of them copying from the giant before), scanned with BVG_DEBUG=1.
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import webgraph_big_amd as W
import tooling as T

def make(n, giants, deg, seed=1):
    rng = np.random.default_rng(seed)
    d = rng.poisson(10, n).astype(np.int64)
    gpos = np.sort(rng.choice(np.arange(100, n - 100), giants, replace=False))
    # giants in adjacent pairs, so that the second can copy from the first
    for i in range(0, giants - 1, 2): gpos[i + 1] = gpos[i] + 1
    lists = {}
    prev = None
    for i, g in enumerate(gpos):
        if i % 2 == 1 and prev is not None:
            keep = prev[rng.random(prev.size) < 0.8]
            extra = rng.choice(n, deg // 5, replace=False)
            l = np.union1d(keep, extra)
        else:
            l = np.sort(rng.choice(n, deg, replace=False))
        lists[int(g)] = l.astype(np.int64); prev = l
    for g in lists: d[g] = lists[g].size
    off = np.zeros(n + 1, np.int64); np.cumsum(d, out=off[1:])
    adj = np.empty(off[-1], np.int64)
    for x in range(n):
        if x in lists: adj[off[x]:off[x + 1]] = lists[x]
        else:
            k = d[x]
            if k: adj[off[x]:off[x + 1]] = np.sort(rng.choice(np.arange(max(0, x - 5000), min(n, x + 5000)), k, replace=False))
    return off, adj

if __name__ == "__main__":
    n = int(os.environ.get("N", 200000)); giants = int(os.environ.get("G", 16)); deg = int(os.environ.get("D", 1000000))
    deg = min(deg, n // 2)
    t0 = time.time(); off, adj = make(n, giants, deg); print("adjacency: %d nodes %d arcs, %.1f s" % (n, adj.size, time.time() - t0), flush=True)
    t0 = time.time(); st = T.store((off.astype(np.uint64), adj), threads=16); print("stored: %.1f MB, %.1f s" % (st.graph.nbytes / 1e6, time.time() - t0), flush=True)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    for it in range(3):
        t0 = time.time(); r = g.scan(); dt = time.time() - t0
        print("scan %d: %.3f s wall, kernel %.1f ms, %d arcs, %.2f G edges/s, slow_blocks %d" % (it, dt, r["kernel_ms"], r["arcs"], r["arcs"] / dt / 1e9, r["slow_blocks"]), flush=True)
    assert r["arcs"] == adj.size
