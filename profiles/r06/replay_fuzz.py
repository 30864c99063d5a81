"""Replays tests/test_gpu_fuzz.py from a given case of a seed (the generator is advanced without running the cases before it) and keeps the failing graph:\n    python profiles/r06/replay_fuzz.py <seed> <first case> <last case>   -> gpurun_out/fail_{off,adj}.npy, fail_case.pkl"""
import os, sys, pickle, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ["BVG_TEST_KNOBS"] = "1"
import webgraph_big_amd as W, tooling as tools
from oracle import bvg_oracle as oracle
from test_gpu_fuzz import _adjacency, _oracle_graph
seed, first, last = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
statef = os.path.join(ROOT, "profiles", "r06", "fuzz_state_%d_%d.pkl") % (seed, first)       # (the generator state in front of case `first`, saved by an earlier replay)
rng = np.random.default_rng(seed)
start = 0
if os.path.exists(statef):
    rng.bit_generator.state = pickle.load(open(statef, "rb")); start = first
for case in range(start, last):
    for k in ("BVG_GIANT", "BVG_NOSKIP", "BVG_EMIT", "BVG_DBG"): os.environ.pop(k, None)
    if case == first and start == 0: pickle.dump(rng.bit_generator.state, open(os.path.join(ROOT, "gpurun_out", os.path.basename(statef)), "wb"))
    n = int(rng.choice([1, 70, 900, 6000, 45000]))
    kw = dict(window_size=int(rng.choice([0, 1, 3, 7, 20, 70])), max_ref_count=int(rng.choice([0, 1, 3, 50, -1])), min_interval_length=int(rng.choice([0, 2, 4, 7])), zeta_k=int(rng.choice([1, 2, 3, 5])))
    if rng.random() < 0.3:
        kw.update(outdegree_coding=int(rng.choice([1, 2])), block_coding=int(rng.choice([1, 2, 5])), residual_coding=int(rng.choice([1, 2, 3, 6, 7])), reference_coding=int(rng.choice([1, 2, 5])), block_count_coding=int(rng.choice([1, 2, 5])))
        if kw["residual_coding"] == 3: kw["zeta_k"] = int(rng.choice([1, 3, 5, 8]))
    tier = str(rng.choice(["default", "giant", "giant", "tasks", "pipelined", "generic"]))
    env = {"giant": dict(BVG_GIANT="2"), "tasks": dict(BVG_EMIT="1", BVG_DBG="16"), "pipelined": dict(BVG_EMIT="0")}.get(tier, {})
    if rng.random() < 0.3: env["BVG_NOSKIP"] = "1"
    off, adj = _adjacency(rng, n)
    p = W.default_params(**kw)
    wide = bool(rng.random() < 0.2)
    a, b = sorted(int(v) for v in rng.integers(0, n + 1, 2)); rng.integers(0, n, min(n, 40))
    if n <= 6000 and adj.size: rng.random()
    rng.choice([0, 64, 1000])
    if case < first: continue
    for k, v in env.items(): os.environ[k] = v
    st = tools.store((off, adj), p, threads=2)
    what = (case, n, kw, tier, env, wide)
    try:
        g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
        if tier == "generic" or wide: g.set_tuning(force_slow=tier == "generic", force_wide=wide)
        og = _oracle_graph(oracle, st); o = og.scan()
        for i in range(2):
            r = g.scan()
            assert (r["nodes"], r["arcs"], r["chk"]) == (o["nodes"], o["arcs"], o["chk"]), (i, r, o)
        ra, oa = g.scan(a, b), og.scan(a, b)
        assert (ra["arcs"], ra["chk"]) == (oa["arcs"], oa["chk"])
        g.close()
        print("case ok", what, flush=True)
    except Exception as ex:
        print("CASE FAILED", what, repr(ex), flush=True)
        np.save(os.path.join(ROOT, "gpurun_out", "fail_off.npy"), off); np.save(os.path.join(ROOT, "gpurun_out", "fail_adj.npy"), adj); pickle.dump((kw, tier, env, wide, n), open(os.path.join(ROOT, "gpurun_out", "fail_case.pkl"), "wb"))
        break
