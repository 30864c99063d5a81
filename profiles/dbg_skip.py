import numpy as np, sys, os
sys.path.insert(0,'.')
import webgraph_big_amd as W
from webgraph_big_amd import tools as T
from oracle import bvg_oracle as O
n = 5000
st = T.synth_store(n, seed=2, chunk_nodes=1024, threads=2)
og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
deg0, succ0 = og.decode_range(0, n)
cum = np.concatenate([[0], np.cumsum(deg0)])
g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
r = g.scan(); ro = og.scan()
print('scan parity', r['chk'] == ro['chk'], r['arcs'], ro['arcs'])
deg, succ = g.decode_range(0, n)
print('deg equal', np.array_equal(deg[:n], deg0))
bad = [x for x in range(n) if not np.array_equal(succ[cum[x]:cum[x+1]], succ0[cum[x]:cum[x+1]])]
print('bad nodes', len(bad), bad[:20])
for x in bad[:3]:
    a = succ[cum[x]:cum[x+1]]; b = succ0[cum[x]:cum[x+1]]
    k = np.nonzero(a != b)[0]
    print(x, 'd', len(b), 'first diff at', k[:5], a[k[:5]], b[k[:5]])
