import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import webgraph_big_amd as W
import tooling as T
st = T.synth_store(1 << 21, seed=0, params=W.default_params(), synth=T.eu_like(mean_deg=127.5), threads=16)
base = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
g = base.tile(36)
for i in range(6):
    t0 = time.perf_counter(); r = g.scan(); dt = time.perf_counter() - t0
    print("scan %d: wall %.1f ms kernel %.1f ms launches %d slow %d" % (i, dt * 1e3, r["kernel_ms"], r["launches"], r["slow_blocks"]))
