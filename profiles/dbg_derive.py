import numpy as np, sys
sys.path.insert(0,'.')
import webgraph_big_amd as W
from webgraph_big_amd import tools as T
from oracle import bvg_oracle as O
for n in (50, 300, 2000, 6000):
  for kw in (dict(block_count_coding=5),):
    st = T.synth_store(n, seed=17, params=W.default_params(**kw), chunk_nodes=1024, threads=2)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    print(n, 'rows parity', g.scan()['chk'] == og.scan()['chk'])
    try:
        h = W.BVGraph.from_memory(st.params, st.graph, None)
        off = h.offsets(); bad = np.nonzero(off != st.offsets)[0]
        print(n, kw, 'ok' if len(bad)==0 else ('first mismatch at %d' % bad[0]))
    except Exception as e:
        print(n, kw, 'EXC', type(e).__name__, e)
