"""Debug: which entries differ between the dense-walk build and the one-pass build of the skip index (heavy-tailed graph)."""
import os, sys, struct, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BVG_TEST_KNOBS"] = "1"
import numpy as np
import webgraph_big_amd as W
import tooling as T
st = T.synth_store(20000, seed=23, params=W.default_params(), synth=T.eu_like(max_deg=30000, tail_alpha=1.6, mean_deg=40.0), threads=4)
d = tempfile.mkdtemp(); base = os.path.join(d, "g"); st.write(base)
out = {}
for mode in ("0", "1"):
    os.environ["BVG_INDEX_WALK"] = mode
    g = W.BVGraph.load(base); g.build_index(); g.scan(); p = os.path.join(d, "i" + mode); g.save_index(p); g.close()
    b = open(p, "rb").read()
    nblk = struct.unpack_from("<I", b, 64 + 8)[0]   # window, wide, nblk ...
    hdr = 128
    magic, version, block_bits, gbytes, tbits, nodes, shash, window, wide, nblk, has_skip, slo, shi, stotal = struct.unpack_from("<8sIIQQqQIIIIIIQ", b, 0)
    o = hdr
    first = np.frombuffer(b, np.uint64, nblk + 1, o); o += (nblk + 1) * 8
    maxd = np.frombuffer(b, np.uint32, nblk, o); o += nblk * 4
    halo = np.frombuffer(b, np.uint32, nblk, o); o += nblk * 4
    mask = np.frombuffer(b, np.uint64, nblk, o); o += nblk * 8
    sfirst = np.frombuffer(b, np.uint64, nblk + 1, o); o += (nblk + 1) * 8
    fmt = np.frombuffer(b, np.uint8, nblk, o); o += nblk
    bit = np.frombuffer(b, np.uint16, stotal, o); o += stotal * 2
    val = np.frombuffer(b, np.uint32, stotal, o); o += stotal * 4
    out[mode] = dict(first=first, maxd=maxd, halo=halo, mask=mask, sfirst=sfirst, fmt=fmt, bit=bit, val=val)
    print(mode, "nblk", nblk, "entries", stotal, "fmt counts", np.bincount(fmt, minlength=4))
a, c = out["0"], out["1"]
for k in a:
    same = np.array_equal(a[k], c[k])
    print(k, "same" if same else "DIFFERENT")
    if not same and k in ("bit", "val"):
        idx = np.nonzero(a[k] != c[k])[0]
        blk = np.searchsorted(a["sfirst"], idx, side="right") - 1
        ub = np.unique(blk)
        print("  ", len(idx), "entries in", len(ub), "blocks; fmt (one-pass / walk) of the first:", [(int(b_), int(a["fmt"][b_]), int(c["fmt"][b_]), int(a["sfirst"][b_ + 1] - a["sfirst"][b_])) for b_ in ub[:10]])
        for i in idx[:8]:
            print("   entry", int(i), "block", int(np.searchsorted(a["sfirst"], i, side="right") - 1), "one-pass", int(a["bit"][i]), int(a["val"][i]), "walk", int(c["bit"][i]), int(c["val"][i]))
    if not same and k == "fmt":
        idx = np.nonzero(a[k] != c[k])[0]; print("  blocks", idx[:10], a[k][idx[:10]], c[k][idx[:10]])
