"""One-off randomised stress of the decode paths against the CPU oracle (a longer version of
tests/test_gpu_parity.py::test_randomised_parameters_and_shapes): parameters x codings x shapes x sizes x kernel variants.
usage: python profiles/stress.py [trials] [seed]"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import webgraph_big_amd as W
import tooling as T
from oracle import bvg_oracle as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time(); bad = 0
for trial in range(trials):
    kw = dict(window_size=int(rng.choice([0, 1, 2, 7, 16, 64, 65, 130])), max_ref_count=int(rng.choice([0, 1, 3, 1000])),
              min_interval_length=int(rng.choice([0, 1, 2, 4, 9])), zeta_k=int(rng.choice([1, 2, 3, 4, 7])))
    if rng.random() < 0.3:
        kw.update(outdegree_coding=int(rng.choice([1, 2])), block_coding=int(rng.choice([1, 2, 5])), residual_coding=int(rng.choice([1, 2, 3, 6, 7])),
                  reference_coding=int(rng.choice([1, 2, 5])), block_count_coding=int(rng.choice([1, 2, 5])))
        if kw["residual_coding"] == 3:
            kw["zeta_k"] = int(rng.choice([1, 3, 5, 8]))
    n = int(rng.choice([1, 2, 63, 64, 65, 500, 3000, 20000, 60000]))
    synth = T.web_like(mean_deg=float(rng.choice([2, 10, 60, 150])), p_copy=float(rng.choice([0.0, 0.5, 0.95])), p_empty=float(rng.choice([0.0, 0.3, 0.9])),
                       p_interval=float(rng.choice([0.0, 0.5, 0.9])), max_deg=int(rng.choice([5, 300, 5000, 30000])), window=int(rng.choice([1, 7, 30, 120])),
                       extra_mean=float(rng.choice([0.5, 4, 30])), keep_run=float(rng.choice([1.5, 12, 40])), skip_run=float(rng.choice([1.2, 3])))
    st = T.synth_store(n, seed=int(rng.integers(1 << 30)), params=W.default_params(**kw), synth=synth, chunk_nodes=int(rng.choice([64, 1 << 16])), threads=4)
    env = {}
    mode = int(rng.integers(0, 6))
    if mode == 1: env = dict(BVG_WG="2", BVG_EMIT="1")
    elif mode == 2: env = dict(BVG_WG="4", BVG_EMIT="1")
    elif mode == 3: env = dict(BVG_EMIT="1", BVG_DBG="16")
    elif mode == 4: env = dict(BVG_EMIT="0")
    elif mode == 5: env = dict(BVG_NOSKIP="1", BVG_EMIT="1")
    for k in ("BVG_WG", "BVG_EMIT", "BVG_DBG", "BVG_NOSKIP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
    ok = True
    try:
        odeg, osucc = og.decode_range(0, n); o = og.scan()
        for rep in range(2):
            r = g.scan()
            ok &= (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"])
        deg, succ = g.decode_range(0, n)
        ok &= bool(np.array_equal(deg, odeg) and np.array_equal(succ, osucc))
        a = int(rng.integers(0, n)); b = int(rng.integers(a, n + 1))
        ra, oa = g.scan(a, b), og.scan(a, b)
        ok &= (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"])
    except Exception as e:
        ok = False; print("EXC", repr(e))
    if not ok:
        bad += 1; print("MISMATCH trial", trial, kw, n, env, {f: getattr(synth, f) for f, _ in synth._fields_ if f != "pad"})
    g.close()
    if trial % 50 == 49:
        print("trial", trial + 1, "bad", bad, "%.0fs" % (time.time() - t0), flush=True)
print("done: %d trials, %d mismatches, %.0f s" % (trials, bad, time.time() - t0))
sys.exit(1 if bad else 0)
