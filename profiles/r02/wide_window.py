import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import webgraph_big_amd as W
import tooling as T
for w, mr in ((7, 3), (64, 3), (70, 3), (70, 1000)):
    st = T.synth_store(1 << 20, seed=3, params=W.default_params(window_size=w, max_ref_count=mr), synth=T.eu_like(), threads=16)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    g.scan()
    t0 = time.time(); r = g.scan(); dt = time.time() - t0
    print("window %d maxref %d: %.1f MB, %.2f bits/arc, scan %.1f ms = %.2f G edges/s (slow_blocks %d)" % (w, mr, st.graph.nbytes / 1e6, 8.0 * st.graph.nbytes / r["arcs"], dt * 1e3, r["arcs"] / dt / 1e9, r["slow_blocks"]), flush=True)
    g.close()
