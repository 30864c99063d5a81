#!/usr/bin/env python3
"""Parity of the flow scan kernel (BVG_FLOW=1) against the CPU oracle on several shapes, then its rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["BVG_FLOW"] = "1"
import numpy as np
import webgraph_big_amd as W
import tooling as T
from oracle import bvg_oracle as O
ok = True
for name, n, sp, kw in (("eu", 30000, T.eu_like(), {}), ("eu15", 30000, T.eu_like(mean_deg=127.5), {}), ("web", 50000, T.web_like(), {}),
                        ("w0", 40000, T.web_like(), dict(window_size=0, max_ref_count=0, min_interval_length=0)),
                        ("tail", 12000, T.eu_like(max_deg=30000, tail_alpha=1.6, mean_deg=40.0), {}), ("minint2", 20000, T.eu_like(), dict(min_interval_length=2, zeta_k=5))):
    st = T.synth_store(n, seed=11, params=W.default_params(**kw), synth=sp, threads=4)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    o = og.scan()
    for rnd in range(3):
        r = g.scan()
        good = (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])
        print("%-8s scan %d: %s  slow_blocks %d launches %d kernel %.2f ms" % (name, rnd, "OK" if good else "MISMATCH arcs %d/%d chk %x/%x" % (r["arcs"], o["arcs"], r["chk"], o["chk"]), r["slow_blocks"], r["launches"], r["kernel_ms"]))
        ok &= good
    for a, b in ((0, 1), (n // 3, n - 7), (4097, 4099)):
        r = g.scan(a, b); oo = og.scan(a, b)
        good = (r["arcs"], r["chk"]) == (oo["arcs"], oo["chk"]); ok &= good
        if not good: print("   range", a, b, "MISMATCH")
    g.close()
print("ALL OK" if ok else "FAILURES")
