"""bvg_tuning.no_index = 2, "marks only" (include/bvgraph_hip.h): the index a handle builds keeps the validation marks -- the lean scan kernel takes the blocks -- and skip
entries only for lists of 4 096 residuals and more.  Same successors, same checksum as the fully indexed scan and the oracle; far fewer entries."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_marks_only_index_scans_with_the_lean_kernel(W, tools, oracle):
    st = tools.synth_store(30000, seed=11, synth=tools.eu_like(mean_deg=60.0), threads=4)
    og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    o = og.scan()
    full = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    lean = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    lean.set_tuning(no_index=2)
    rf = rl = None
    for _ in range(3):
        rf, rl = full.scan(), lean.scan()
        assert (rf["arcs"], rf["chk"]) == (o["arcs"], o["chk"]) and (rl["arcs"], rl["chk"]) == (o["arcs"], o["chk"])
    assert rl["lean_blocks"] > 0 and rl["lean_blocks"] == rf["lean_blocks"]          # the same blocks are validated
    assert rf["index_entries"] > 100 * max(rl["index_entries"], 1)                     # ... with next to no entries
    deg, succ = lean.decode_range(0, lean.num_nodes())
    odeg, osucc = og.decode_range(0, lean.num_nodes())
    assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc)
    a, b = 7001, 19001
    assert lean.scan(a, b)["chk"] == og.scan(a, b)["chk"]
    full.close(); lean.close()
