"""TEST INFRASTRUCTURE (emulated library): a device allocation fails while the first big scan builds the residual skip index.  The scan must still be exact (index-less),
say so once on stderr without BVG_DEBUG, NOT pay the counting pass on every later scan, and build the index when it tries again (every 8th scan) -- ADVICE r4, bvg_api.hip give_up()."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ["BVG_HIP_LIB"] = os.path.join(HERE, "libbvgraph_emu.so")
os.environ["BVG_TEST_KNOBS"] = "1"
os.environ.pop("BVG_DEBUG", None)

import tooling as T  # noqa: E402
import webgraph_big_amd as W  # noqa: E402
from oracle import bvg_oracle as O  # noqa: E402

st = T.synth_store(9000, seed=4, synth=T.web_like(), threads=2)
g = W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=0)
og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
o = og.scan()
r = g.scan(0, 1000); assert r["chk"] == og.scan(0, 1000)["chk"]          # builds the block plan; too short a scan to build the index
emu = ctypes.CDLL(os.environ["BVG_HIP_LIB"])
emu.emu_fail_next_mallocs(1)
lean = []
for i in range(10):
    r = g.scan()
    assert (r["arcs"], r["chk"]) == (o["arcs"], o["chk"]), (i, r, o)
    lean.append(int(r["lean_blocks"]))
print("lean blocks per scan:", lean)
assert lean[0] == 0 and all(v == 0 for v in lean[:7]), lean                 # index-less while the failure is remembered ...
assert lean[-1] > 0, lean                                                   # ... and indexed once the retry went through
print("oom case ok")
