"""Materialise-mode throughput: bvg_decode_range_dev of a node range into device buffers (the NodeIterator batch path)."""
import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import webgraph_big_amd as W
import tooling as T
shape = sys.argv[1] if len(sys.argv) > 1 else 'eu'
st = T.synth_store(1 << 21, seed=0, synth=T.eu_like() if shape == 'eu' else T.web_like(), threads=32)
base = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
K = 8
g = base.tile(K)
n = g.num_nodes(); arcs = st.stats['arcs'] * K
d_deg = torch.empty(n, dtype=torch.int32, device='cuda'); d_succ = torch.empty(arcs, dtype=torch.int64, device='cuda')
need = C.c_uint64()
L = W.lib()
times = []
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st_ = L.bvg_decode_range_dev(g._h, 0, n, d_deg.data_ptr(), d_succ.data_ptr(), arcs, C.byref(need))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert st_ == 0 and need.value == arcs, (st_, need.value, arcs)
    times.append(dt)
dt = min(times[2:])                                                   # (the first call builds the index, the second learns the tiers)
print(shape, 'materialise %d arcs in %.1f ms -> %.1f G edges/s, %.1f GB/s written' % (arcs, dt * 1e3, arcs / dt / 1e9, arcs * 8 / dt / 1e9))
# spot check vs first tile through the host path
deg, succ = base.decode_range(0, 1000)
assert torch.equal(d_succ[:len(succ)].cpu(), torch.from_numpy(succ)) and torch.equal(d_deg[:1000].cpu(), torch.from_numpy(deg))
print('spot check ok')
import json, os
print('JSON ' + json.dumps({'shape': shape, 'arcs': int(arcs), 'nodes': int(n), 'seconds': dt, 'edges_per_s': arcs / dt, 'bytes_written_per_s': arcs * 8 / dt, 'scan_kernel': os.environ.get('BVG_SCANK', '1') != '0', 'all_calls_s': times}))
