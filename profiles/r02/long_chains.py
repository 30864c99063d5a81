# r02: reference chains without a bound (maxrefcount large): a block boundary is cuttable only if the chains of the nodes behind it stay
# within 64 nodes -- how many boundaries survive, and what does the scan rate become?
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import webgraph_big_amd as W
import tooling as T
for mr in (3, 10, 100, -1):
    st = T.synth_store(1 << 21, seed=3, params=W.default_params(window_size=7, max_ref_count=mr), synth=T.eu_like(), threads=16)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    g.scan(); g.scan()
    t0 = time.time(); r = g.scan(); dt = time.time() - t0
    print("maxrefcount %d: %.1f MB, %.2f bits/arc, scan %.1f ms = %.2f G edges/s, %d launches, slow_blocks %d" % (mr, st.graph.nbytes / 1e6, 8.0 * st.graph.nbytes / r["arcs"], dt * 1e3, r["arcs"] / dt / 1e9, r["launches"], r["slow_blocks"]), flush=True)
    g.close()
