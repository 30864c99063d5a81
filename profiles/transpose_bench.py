"""Rate of the transposition feed (decode + device sort) on the eu- and web-shaped workloads; run on the GPU box."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import webgraph_big_amd as W
import tooling as T
for shape, synth in (("eu", T.eu_like()), ("web", T.web_like())):
    st = T.synth_store(1 << 21, seed=0, params=W.default_params(), synth=synth, threads=32)
    base = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    copies = int(float(sys.argv[1]) * (1 << 30) / len(st.graph)) if len(sys.argv) > 1 else 4
    g = base.tile(copies) if copies > 1 else base
    n, m = g.num_nodes(), st.stats["arcs"] * copies
    d_off = torch.empty(n + 1, dtype=torch.int64, device="cuda"); d_ts = torch.empty(m, dtype=torch.int64, device="cuda")
    import ctypes as C
    need = C.c_uint64(0)
    L = W.bvgraph.lib() if hasattr(W, "bvgraph") else None
    from importlib import import_module
    L = import_module("webgraph-big_amd.bvgraph").lib()
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = L.bvg_transpose_dev(g._h, d_off.data_ptr(), d_ts.data_ptr(), m, C.byref(need))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert rc == 0 and need.value == m, (rc, need.value, m)
    assert int(d_off[-1].item()) == m and bool((d_off[1:] >= d_off[:-1]).all())
    print("%s: %d nodes, %.2f G arcs: transpose feed %.1f ms = %.2f G arcs/s (decode + stable 64-bit radix sort of %d key bits + in-degree prefix)"
          % (shape, n, m / 1e9, dt * 1e3, m / dt / 1e9, int(np.ceil(np.log2(max(n, 2))))))
    d_so = torch.empty(n + 1, dtype=torch.int64, device="cuda"); d_ss = torch.empty(2 * m, dtype=torch.int64, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = L.bvg_symmetrize_dev(g._h, d_so.data_ptr(), d_ss.data_ptr(), 2 * m, C.byref(need))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert rc == 0 and m <= need.value <= 2 * m, (rc, need.value, m)
    print("%s: symmetrise (transpose feed + per-node union): %.2f G arcs out in %.1f ms = %.2f G input arcs/s" % (shape, need.value / 1e9, dt * 1e3, m / dt / 1e9))
