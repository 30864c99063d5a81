"""Label co-decode throughput: bvg_decode_range_dev followed by bvg_labels_decode_range_dev on device buffers (the labelled
node-iterator batch, labelling/BitStreamArcLabelledImmutableGraph.java:565-582), gamma-coded and 10-bit labels."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import webgraph_big_amd as W
import tooling as T
shape = sys.argv[1] if len(sys.argv) > 1 else 'eu'
n = 1 << 21
synth = T.eu_like() if shape == 'eu' else T.web_like()
st = T.synth_store(n, seed=0, synth=synth, threads=32)
off, adj = T.synth_adjacency(n, seed=0, synth=synth)
g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
arcs = int(off[-1])
rng = np.random.default_rng(0)
L = W.lib()
d_deg = torch.empty(n, dtype=torch.int32, device='cuda'); d_succ = torch.empty(arcs, dtype=torch.int64, device='cuda')
d_lab = torch.empty(arcs, dtype=torch.int32, device='cuda')
need = C.c_uint64()
assert L.bvg_decode_range_dev(g._h, 0, n, d_deg.data_ptr(), d_succ.data_ptr(), arcs, C.byref(need)) == 0
for kind, width, name in ((1, 0, 'gamma'), (2, 10, 'fixed10')):
    vals = (rng.geometric(0.02, size=arcs) - 1).astype(np.int32) if kind == 1 else rng.integers(0, 1024, size=arcs, dtype=np.int32)
    sl = T.store_labels(kind, width, vals, off)
    lg = W.BitStreamArcLabelledImmutableGraph.from_memory(g, kind, width, sl.stream, sl.offsets)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = L.bvg_labels_decode_range_dev(lg._h, 0, n, d_deg.data_ptr(), d_lab.data_ptr(), arcs, C.byref(need))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert r == 0 and need.value == arcs
    assert np.array_equal(d_lab.cpu().numpy(), vals)
    print('%s %s: %d labels (%.2f bits each) in %.2f ms -> %.1f G labels/s, %.1f GB/s of label stream' %
          (shape, name, arcs, len(sl.stream) * 8 / arcs, dt * 1e3, arcs / dt / 1e9, len(sl.stream) / dt / 1e9))
    lg.close()
