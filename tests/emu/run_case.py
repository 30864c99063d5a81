"""TEST INFRASTRUCTURE: one graph through the EMULATED library (tests/emu/libbvgraph_emu.so: the product's kernels compiled for the host,
lanes as fibers) against the CPU oracle.  Run as a child process -- the library is chosen when the package first loads it:
    python tests/emu/run_case.py <nodes> <seed> [shape] [scans]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ["BVG_HIP_LIB"] = os.path.join(HERE, os.environ.get("BVG_EMU_LIB", "libbvgraph_emu.so"))     # (an absolute path wins: tests/test_gpu_flat.py runs the same cases on the GPU)
os.environ.setdefault("BVG_TEST_KNOBS", "1")

import numpy as np  # noqa: E402
import tooling as T  # noqa: E402
import webgraph_big_amd as W  # noqa: E402
from oracle import bvg_oracle as O  # noqa: E402


def main():
    n = int(sys.argv[1]); seed = int(sys.argv[2]); shape = sys.argv[3] if len(sys.argv) > 3 else "web"; scans = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    if shape == "cnr":
        base = os.path.join(ROOT, "tests", "golden", "cnr-2000")
        g = W.BVGraph.load(base, device=0)
        og = O.Graph.load(base)
        n = min(n, g.num_nodes())
    else:
        synth = T.web_like() if shape == "web" else T.eu_like(mean_deg=float(os.environ.get("EMU_DEG", "60")))
        st = T.synth_store(n, seed=seed, synth=synth, threads=4)
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=0)
        og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    t0 = time.time()
    o = og.scan(0, n)
    for i in range(scans):
        r = g.scan(0, n)
        print("scan %d: arcs %d chk %016x lean_blocks %d slow_blocks %d (%.1f s)" % (i, r["arcs"], r["chk"], r.get("lean_blocks", -1), r.get("slow_blocks", -1), time.time() - t0), flush=True)
        assert (r["arcs"], r["chk"], r["nodes"]) == (o["arcs"], o["chk"], o["nodes"]), (r, o)
    lo, hi = n // 3, n // 3 + min(n - n // 3, 3000)
    deg, succ = g.decode_range(lo, hi)
    odeg, osucc = og.decode_range(lo, hi)
    assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc)
    r = g.scan(lo, hi); o2 = og.scan(lo, hi)
    assert (r["arcs"], r["chk"]) == (o2["arcs"], o2["chk"])
    print("emu case ok: %d nodes, lean_blocks %d" % (n, r.get("lean_blocks", -1)))


if __name__ == "__main__":
    main()
