import os,sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import numpy as np
import tooling as T, webgraph_big_amd as W
from oracle import bvg_oracle as O
st = T.synth_store(120000, seed=31, synth=T.eu_like(mean_deg=60.0), threads=8)
g = W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=0)
og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
o = og.scan()
for i in range(3):
    r = g.scan()
    print(os.environ.get('WHAT'), i, (r['arcs'], r['chk']) == (o['arcs'], o['chk']), r['lean_blocks'], r['slow_blocks'])
