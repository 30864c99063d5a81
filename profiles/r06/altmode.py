"""A handle that alternates bvg_scan and bvg_decode_range: what a scan costs right after a materialising call (round 6: the learned tiers are kept per mode).
    python profiles/r06/altmode.py   -> ms per scan: steady | first scan after a decode_range of one tile | the one after"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import webgraph_big_amd as W, tooling as T
st = T.synth_store(1 << 20, seed=3, synth=T.eu_like(), threads=16)
base = W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=0)
g = W.mosaic([base], 24)
n0 = 1 << 20
def scan_ms():
    torch.cuda.synchronize(); t = time.perf_counter(); r = g.scan(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3, r
for _ in range(4): scan_ms()
steady = np.mean([scan_ms()[0] for _ in range(4)])
deg, succ = g.decode_range(5 * n0, 6 * n0); del deg, succ
a, r1 = scan_ms(); b, r2 = scan_ms()
deg, succ = g.decode_range(5 * n0, 6 * n0); del deg, succ
c, r3 = scan_ms()
print("ms per scan: steady %.2f | after a decode_range %.2f, then %.2f | after a second one %.2f | slow blocks %d %d %d" % (steady, a, b, c, r1["slow_blocks"], r2["slow_blocks"], r3["slow_blocks"]))
