"""Diagnostic (GPU box): which nodes does the scan path get wrong?  Scans [x, x+1) for every node and compares the checksum with the
oracle's; prints the record headers of the first mismatches."""
import os, sys
os.environ.setdefault("BVG_TEST_KNOBS", "1"); os.environ.setdefault("BVG_EMIT", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import webgraph_big_amd as W
import tooling as T
from oracle import bvg_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
st = T.synth_store(n, seed=41, synth=T.eu_like(), threads=4)
g = W.BVGraph.from_memory(st.params, st.graph, st.offsets)
og = O.Graph.from_memory(O.Params(**st.params.as_dict()), st.graph.tobytes(), st.offsets)
g.build_index()
r, o = g.scan(), og.scan()
print("whole:", r["chk"] == o["chk"], r["lean_blocks"], r["slow_blocks"])
bits = np.unpackbits(np.frombuffer(st.graph.tobytes(), dtype=np.uint8))


class Rd:
    def __init__(self, pos): self.p = pos
    def unary(self):
        z = 0
        while bits[self.p] == 0: z += 1; self.p += 1
        self.p += 1; return z
    def take(self, k):
        v = 0
        for _ in range(k): v = (v << 1) | int(bits[self.p]); self.p += 1
        return v
    def gamma(self):
        m = self.unary(); return ((1 << m) | self.take(m)) - 1


def header(x, degs):
    rd = Rd(int(st.offsets[x])); d = rd.gamma()
    if d == 0: return dict(d=0)
    ref = rd.unary(); blocks = []; ic = 0; ivl = []
    extra = d
    if ref:
        bc = rd.gamma(); blocks = [rd.gamma() + (1 if i else 0) for i in range(bc)]
        copied = sum(blocks[0::2]); tot = sum(blocks)
        if bc % 2 == 0: copied += degs[x - ref] - tot
        extra = d - copied
    if extra > 0:
        ic = rd.gamma()
        for i in range(ic):
            rd.gamma(); l = rd.gamma() + 4; ivl.append(l); extra -= l
    return dict(d=d, ref=ref, bc=len(blocks), blocks=blocks[:6], ic=ic, ivl=ivl[:4], nres=extra)


degs, _ = og.decode_range(0, n)
bad = []; lean = {}
for x in range(n):
    a, b = g.scan(x, x + 1), og.scan(x, x + 1)
    lean[x] = a["lean_blocks"]
    if a["chk"] != b["chk"] or a["arcs"] != b["arcs"]:
        bad.append(x)
print("mismatching nodes:", len(bad), "of", n, "; of them in lean blocks:", sum(lean[x] for x in bad), "; nodes in lean blocks:", sum(lean.values()))
os.environ["BVG_SCANK"] = "0"
bad0 = [x for x in range(n) if g.scan(x, x + 1)["chk"] != og.scan(x, x + 1)["chk"]]
print("with BVG_SCANK=0 mismatching:", len(bad0))
del os.environ["BVG_SCANK"]
referenced = set()
for x in range(n):
    if degs[x]:
        rd = Rd(int(st.offsets[x])); rd.gamma(); ref = rd.unary()
        if ref: referenced.add(x - ref)
for x in bad[:25]:
    print(x, "lean" if lean[x] else "rows", header(x, degs), "referenced-by-someone" if x in referenced else "leaf")
kinds = {}
for x in bad:
    h = header(x, degs)
    key = ("ref" if h.get("ref") else "noref", "iv" if h.get("ic") else "noiv", "res>=24" if h.get("nres", 0) >= 24 else ("res" if h.get("nres", 0) else "nores"), "referenced" if x in referenced else "leaf")
    kinds[key] = kinds.get(key, 0) + 1
print(kinds)
