"""Offsets derivation from a bare .graph (BVGraph -O): seconds per GiB of stream, rounds, exactness; eu / web / w0 shapes."""
import os, sys, time
os.environ.setdefault("BVG_TEST_KNOBS", "1"); os.environ["BVG_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import webgraph_big_amd as W
import tooling as T
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
for shape, synth, kw in (("eu", T.eu_like(mean_deg=127.5), {}), ("web", T.web_like(), {}), ("w0", T.web_like(), dict(window_size=0, max_ref_count=0, min_interval_length=0))):
    st = T.synth_store(1 << 20, seed=0, params=W.default_params(**kw), synth=synth, threads=16)
    k = max(1, int(gib * (1 << 30) / len(st.graph)))
    big = T.tile_host(st, k)
    for mode in ("parallel", "sequential") if shape == "eu" and gib <= 0.26 else ("parallel",):
        if mode == "sequential": os.environ["BVG_DERIVE_SEQ"] = "1"
        else: os.environ.pop("BVG_DERIVE_SEQ", None)
        t0 = time.time()
        g = W.BVGraph.from_memory(big.params, big.graph, None)
        dt = time.time() - t0
        ok = np.array_equal(g.offsets(), big.offsets)
        g.close()
        t0 = time.time(); g2 = W.BVGraph.from_memory(big.params, big.graph, big.offsets); up = time.time() - t0; g2.close()
        print("%s %s: %.2f GiB, %d nodes: open without offsets %.2f s (with offsets: %.2f s) -> derivation %.2f s per GiB, exact=%s" % (shape, mode, len(big.graph) / 2**30, big.params.nodes, dt, up, (dt - up) / (len(big.graph) / 2**30), ok), flush=True)
