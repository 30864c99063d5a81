# r02: a transposed graph end to end on the device: synthetic eu-shaped graph -> bvg_transpose -> bvg_store (the GPU compressor) ->
# open -> scans.  Transposes have heavy-tailed lists (in-degrees): the giant-list kernel (tier 2a) against the generic kernel.
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import webgraph_big_amd as W
import tooling as T

def scans(g, what, k=3):
    for it in range(k):
        t0 = time.time(); r = g.scan(); dt = time.time() - t0
        print("  %s scan %d: %.1f ms wall, kernel %.1f ms, %.2f G edges/s, slow_blocks %d" % (what, it, dt * 1e3, r["kernel_ms"], r["arcs"] / dt / 1e9, r["slow_blocks"]), flush=True)
    return r

if __name__ == "__main__":
    n = int(os.environ.get("N", 1 << 21)); shape = os.environ.get("SHAPE", "eu")
    synth = T.eu_like() if shape == "eu" else T.web_like()
    t0 = time.time(); st = T.synth_store(n, seed=7, synth=synth, threads=16)
    print("%s graph: %d nodes, %d arcs, %.1f MB (%.1f s)" % (shape, n, st.params.arcs, st.graph.nbytes / 1e6, time.time() - t0), flush=True)
    g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
    t0 = time.time(); toff, tsucc = g.transpose(); print("transpose on the device: %.2f s" % (time.time() - t0), flush=True)
    g.close()
    hubs = int(os.environ.get("HUBS", 0))
    if hubs:
        # the synthetic generator's in-degrees are light-tailed: add Zipf-popular hubs (every node links to hub r with probability
        # 0.5 / r^0.9), as the home pages of a web graph or the celebrities of a social graph are -- their lists in the transpose
        # are random subsets of all nodes: short gaps, many intervals
        rng = np.random.default_rng(11)
        hub_ids = np.sort(rng.choice(n, hubs, replace=False))
        pr = 0.5 / np.arange(1, hubs + 1) ** 0.9
        rng.shuffle(pr)
        indeg0 = np.diff(toff.astype(np.int64))
        lists = []
        for h, q in zip(hub_ids, pr):
            cnt = rng.binomial(n, q)
            src = np.sort(rng.choice(n, cnt, replace=False)) if cnt else np.empty(0, np.int64)
            lists.append(np.union1d(tsucc[int(toff[h]):int(toff[h + 1])], src))
        newdeg = indeg0.copy(); newdeg[hub_ids] = [l.size for l in lists]
        noff = np.zeros(n + 1, np.uint64); np.cumsum(newdeg, out=noff[1:])
        nsucc = np.empty(int(noff[-1]), np.int64)
        # copy the untouched stretches between hubs in bulk
        prev = 0
        for h, l in zip(hub_ids, lists):
            a0, a1 = int(toff[prev]), int(toff[h]); b0 = int(noff[prev])
            nsucc[b0:b0 + (a1 - a0)] = tsucc[a0:a1]
            nsucc[int(noff[h]):int(noff[h + 1])] = l
            prev = h + 1
        a0, a1 = int(toff[prev]), int(toff[n]); b0 = int(noff[prev]); nsucc[b0:b0 + (a1 - a0)] = tsucc[a0:a1]
        toff, tsucc = noff, nsucc
        print("with %d hubs: %d arcs" % (hubs, tsucc.size), flush=True)
    indeg = np.diff(toff.astype(np.int64))
    big = indeg > 12288
    print("in-degrees: max %d, %d lists > 12288 holding %.1f %% of the arcs; > 100000: %d" % (indeg.max(), big.sum(), 100.0 * indeg[big].sum() / max(1, indeg.sum()), (indeg > 100000).sum()), flush=True)
    t0 = time.time(); tg, toffs = W.store((toff, tsucc)); print("store on the device: %.2f s, %.1f MB, %.2f bits/arc" % (time.time() - t0, tg.nbytes / 1e6, 8.0 * tg.nbytes / tsucc.size), flush=True)
    p = W.default_params(nodes=n, arcs=int(tsucc.size))
    h = W.BVGraph.from_memory(p, tg, toffs)
    r1 = scans(h, "giant kernel")
    h.close()
    os.environ["BVG_GIANT"] = "0"
    h = W.BVGraph.from_memory(p, tg, toffs)
    r0 = scans(h, "generic kernel", 2)
    h.close()
    assert (r0["arcs"], r0["chk"]) == (r1["arcs"], r1["chk"]) and r1["arcs"] == tsucc.size, "the two tiers disagree"
    print("same checksum from both tiers: %x" % r1["chk"])
