"""TEST INFRASTRUCTURE (emulated library): randomised parity of the lean scan kernels -- scan_kernel and, with BVG_FLAT=1, the experimental flat kernel -- against the CPU oracle on
graphs large enough for the index to be built (>= 4 096 nodes): random shape (sparse / dense / with large lists), window, maxrefcount (deep stages), minimum interval length, zeta k,
LDS geometry (small pools force sub-rows and compaction), records per super-row, lane order.
    python tests/emu/fuzz_flat.py <cases> <seed>"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import tooling as T, webgraph_big_amd as W
from oracle import bvg_oracle as O
kw = %(kw)r; n = %(n)d; seed = %(seed)d; dense = %(dense)r
synth = T.eu_like(mean_deg=%(deg)f, p_interval=%(piv)f, max_deg=%(maxd)d) if dense else T.web_like(p_interval=%(piv)f, max_deg=%(maxd)d)
st = T.synth_store(n, seed=seed, params=W.default_params(**kw), synth=synth, threads=2)
g = W.BVGraph.from_memory(st.params, st.graph, st.offsets, device=0)
og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
o = og.scan()
lean = 0
for i in range(3):
    r = g.scan(); lean = r["lean_blocks"]
    assert (r["arcs"], r["chk"]) == (o["arcs"], o["chk"]), (i, r, o)
a, b = n // 5, min(n, n // 5 + 4500)
r = g.scan(a, b); o2 = og.scan(a, b)
assert (r["arcs"], r["chk"]) == (o2["arcs"], o2["chk"])
print("ok lean_blocks", lean)
'''


def main():
    import numpy as np
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8; seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed0)
    bad = 0; lean_cases = 0
    for c in range(cases):
        kw = dict(window_size=int(rng.choice([1, 3, 7, 7, 20])), max_ref_count=int(rng.choice([1, 3, 3, 10, -1])), min_interval_length=int(rng.choice([0, 2, 3, 4, 7])), zeta_k=int(rng.choice([2, 3, 3, 5])))
        dense = bool(rng.random() < 0.5)
        sub = dict(root=ROOT, kw=kw, n=int(rng.choice([5000, 7000, 9000])), seed=int(rng.integers(1, 1 << 20)), dense=dense, deg=float(rng.choice([30.0, 60.0, 120.0])),
                   piv=float(rng.choice([0.0, 0.3, 0.7])), maxd=int(rng.choice([300, 3000, 20000])))
        env = dict(os.environ, BVG_HIP_LIB=os.path.join(HERE, "libbvgraph_emu.so"), BVG_TEST_KNOBS="1", BVG_FLAT=str(int(rng.random() < 0.25)), BVG_FLAT_RECS=str(int(rng.choice([64, 128, 256]))),
                   BVG_EMU_ORDER=str(rng.choice(["fwd", "rev"])))
        # scan_kernel's list builds (BVG_DBG): 0 = position tasks, 8192 = ZE (kept-element tasks over the extras' bit vectors), 4096 = WW (experimental wave-wide build), both
        lb = int(rng.choice([0, 0, 0, 0, 8192, 4096, 12288])) if env["BVG_FLAT"] == "0" or os.environ.get("BVG_FUZZ_DBG") else 0
        if os.environ.get("BVG_FUZZ_DBG"):
            lb = int(os.environ["BVG_FUZZ_DBG"]); env["BVG_FLAT"] = "0"
        if lb:
            env["BVG_DBG"] = str(lb)
        if rng.random() < 0.5:
            env["BVG_SCAN_POOL"] = str(int(rng.choice([512, 640, 1024]))); env["BVG_SCAN_SCR"] = str(int(rng.choice([192, 320, 448])))
        p = subprocess.run([sys.executable, "-c", CHILD % sub], env=env, capture_output=True, text=True, timeout=1200)
        ok = p.returncode == 0 and "ok lean_blocks" in p.stdout
        lean_cases += int(ok and not p.stdout.strip().endswith(" 0"))
        print("case %d %s flat=%s recs=%s order=%s pool=%s dbg=%s %s n=%d dense=%s: %s" % (c, "ok" if ok else "FAILED", env["BVG_FLAT"], env["BVG_FLAT_RECS"], env["BVG_EMU_ORDER"], env.get("BVG_SCAN_POOL", "-"), env.get("BVG_DBG", "0"), kw, sub["n"], dense,
                                                                              p.stdout.strip()[-40:] if ok else (p.stdout[-300:] + p.stderr[-1500:])), flush=True)
        bad += int(not ok)
    print("emu fuzz: %d cases, %d failed, %d ran the lean kernels" % (cases, bad, lean_cases))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
