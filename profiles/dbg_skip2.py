import numpy as np, sys, os
sys.path.insert(0,'.')
import webgraph_big_amd as W
from webgraph_big_amd import tools as T
gib = float(sys.argv[1]); build_mode = sys.argv[2]; use_modes = sys.argv[3].split(',')
st = T.synth_store(1 << 21, seed=0, params=W.default_params(), synth=T.eu_like(), threads=32)
copies = int(gib * (1 << 30) / len(st.graph))
base = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
g = base.tile(copies)
n0 = st.params.nodes
os.environ['BVG_NOSKIP'] = '1'; os.environ['BVG_DBG'] = '0'
ref = g.scan()
print('ref arcs', ref['arcs'], 'chk %x' % ref['chk'])
del os.environ['BVG_NOSKIP']
os.environ['BVG_DBG'] = build_mode
try:
    r = g.scan(); print('build+scan mode', build_mode, 'ok', r['chk'] == ref['chk'])
except Exception as e:
    print('build+scan mode', build_mode, 'EXC', e)
for m in use_modes:
    os.environ['BVG_DBG'] = m
    try:
        r = g.scan(); print('  use mode', m, 'ok', r['chk'] == ref['chk'])
    except Exception as e:
        print('  use mode', m, 'EXC', str(e)[:60])
