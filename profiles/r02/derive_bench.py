#!/usr/bin/env python3
"""Offsets derivation from a bare .graph (BVGraph -O, BVGraph.java:2595-2609): chunk-parallel walk vs the one-wavefront walk.
usage: derive_bench.py [gib]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import webgraph_big_amd as W
import tooling as T
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
st = T.synth_store(1 << 21, seed=0, synth=T.eu_like(), threads=16)
big = T.tile_host(st, max(1, int(gib * (1 << 30) / len(st.graph))))
os.environ["BVG_DERIVE_PAR"] = "1"
t0 = time.time(); g = W.BVGraph.from_memory(big.params, big.graph, None); t1 = time.time() - t0
ok = np.array_equal(g.offsets(), big.offsets); g.close()
print("parallel: %.2f GiB, %d nodes, %d arcs: open+derive %.2f s, offsets equal the encoder's: %s" % (len(big.graph) / 2**30, big.params.nodes, big.stats["arcs"], t1, ok))
t0 = time.time(); g = W.BVGraph.from_memory(big.params, big.graph, big.offsets); t2 = time.time() - t0; g.close()
print("  (open with offsets given: %.2f s, so the derivation itself took about %.2f s)" % (t2, t1 - t2))
if os.environ.get("SEQ"):
    small = T.tile_host(st, 2)
    os.environ.pop("BVG_DERIVE_PAR", None)
    t0 = time.time(); g = W.BVGraph.from_memory(small.params, small.graph, None); t3 = time.time() - t0
    ok = np.array_equal(g.offsets(), small.offsets); g.close()
    print("one wavefront: %.3f GiB: %.2f s (%s) -> %.1f s per GiB" % (len(small.graph) / 2**30, t3, ok, t3 / (len(small.graph) / 2**30)))
