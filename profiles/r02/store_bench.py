#!/usr/bin/env python3
"""BVGraph.store on the device (bvg_store) against the CPU tooling encoder: arcs/s end to end (host adjacency in, host stream out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import webgraph_big_amd as W
import tooling as T
for name, sp, n in (("eu", T.eu_like(), 1 << 21), ("web", T.web_like(), 1 << 22)):
    off, adj = T.synth_adjacency(n, seed=0, synth=sp, chunk_nodes=1 << 16)
    p = W.default_params()
    W.store((off[:1001], adj[:int(off[1000])][adj[:int(off[1000])] < 1000] if False else adj[:0]), p) if False else None
    t0 = time.time(); g1, o1 = W.store((off, adj), p, chunk_nodes=1 << 16); t1 = time.time() - t0
    t0 = time.time(); g2, o2 = W.store((off, adj), p, chunk_nodes=1 << 16); t2 = time.time() - t0
    t0 = time.time(); want = T.store((off, adj), p, chunk_nodes=1 << 16, threads=16); t3 = time.time() - t0
    same = np.array_equal(o2, want.offsets) and g2.tobytes() == want.graph.tobytes()
    print("%-4s %d nodes %d arcs: device store %.2f s (first call %.2f s) = %.1f M arcs/s; CPU tooling on 16 threads %.2f s = %.1f M arcs/s; identical bytes: %s; %.2f bits/arc"
          % (name, n, len(adj), t2, t1, len(adj) / t2 / 1e6, t3, len(adj) / t3 / 1e6, same, 8.0 * len(g2) / len(adj)))
