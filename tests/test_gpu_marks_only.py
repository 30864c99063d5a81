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


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["residuals_only", "dense", "sparse"])
def test_long_lists_without_entries_are_decoded_by_the_whole_wavefront(W, tools, oracle, kind):
    """Marks only: lists of more than 32 residuals have no skip entries; behind their first 32 residuals the wavefront decodes them together, 64 bit positions of the stream per
    step (bvg_scan.hip, "COOP": every lane decodes the code that would start at its bit, a scalar walk over the lengths marks the real starts, a wave prefix sum gives the values).
    Leaves (window 0: nothing is copied), stored lists with and without reference, lists merged straight into the output of decode_range."""
    if kind == "residuals_only": st = tools.synth_store(12000, seed=21, params=W.default_params(window_size=0, max_ref_count=0), synth=tools.eu_like(mean_deg=90.0), threads=4)
    elif kind == "dense": st = tools.synth_store(20000, seed=22, synth=tools.eu_like(mean_deg=120.0), threads=4)
    else: st = tools.synth_store(30000, seed=23, synth=tools.web_like(), threads=4)
    og = oracle.Graph.from_memory(oracle.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    o = og.scan()
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    g.set_tuning(no_index=2)
    for _ in range(3):
        r = g.scan()
        assert (r["arcs"], r["chk"]) == (o["arcs"], o["chk"])
    assert r["lean_blocks"] > 0
    n = g.num_nodes()
    deg, succ = g.decode_range(0, n)
    odeg, osucc = og.decode_range(0, n)
    assert np.array_equal(deg, odeg) and np.array_equal(succ, osucc)
    for a, b in [(0, 1), (n // 3, n // 2), (n - 5, n)]:
        assert g.scan(a, b)["chk"] == og.scan(a, b)["chk"]
    g.close()
