"""CPU tooling (tooling/lib/libbvg_tools.so): the BVGraph *encoder* and the synthetic graph generators that tests and bench.py
manufacture their inputs with.  Test and benchmark infrastructure: outside the product package, never on the GPU hot path, and
held to the oracle's standard (it regenerates the reference's cnr-2000 fixture byte for byte, tests/test_store.py).
Restates BVGraph.java:1595-1618, 1977-2159, 2216-2327 (see tooling/bvg_store.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

import webgraph_big_amd as _W        # (the ABI structs of the product package: the one definition of bvg_params)

Params, default_params, GAMMA = _W.Params, _W.default_params, _W.GAMMA

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("arcs", "copied", "intervalised", "residual", "tot_ref", "tot_dist",
                                          "nodes_with_ref", "graph_bits", "graph_bytes")]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SynthParams(C.Structure):
    _fields_ = [("p_empty", C.c_double), ("mean_deg", C.c_double), ("tail_alpha", C.c_double), ("max_deg", C.c_int64),
                ("p_copy", C.c_double), ("keep_run", C.c_double), ("skip_run", C.c_double), ("p_interval", C.c_double),
                ("interval_len", C.c_double), ("local_gap", C.c_double), ("p_far", C.c_double), ("window", C.c_int32),
                ("pad", C.c_int32), ("extra_mean", C.c_double)]


def web_like(**kw):
    """Copy-model parameters tuned towards cnr-2000's arc provenance."""
    sp = SynthParams(p_empty=0.2, mean_deg=16.0, tail_alpha=2.5, max_deg=3000, p_copy=0.9, keep_run=12.0, skip_run=1.5,
                     p_interval=0.4, interval_len=8.0, local_gap=5.0, p_far=0.04, window=7, pad=0, extra_mean=1.5)
    for k, v in kw.items():
        setattr(sp, k, v)
    return sp


def eu_like(**kw):
    """Denser, more copy-heavy variant (eu-2015-shaped: average outdegree ~60-90, ~2.5 bits/link)."""
    return web_like(**dict(dict(p_copy=0.88, keep_run=25.0, skip_run=2.0, extra_mean=4.0, mean_deg=90.0, p_interval=0.5,
                                interval_len=20.0, max_deg=20000, local_gap=6.0, p_far=0.03, tail_alpha=2.7), **kw))


def build(force=False):
    if os.environ.get("BVG_TOOLS_LIB"):                      # an alternative build of the same source (tooling/Makefile `asan`: tests/test_sanitizers.py)
        return os.environ["BVG_TOOLS_LIB"]
    so = os.path.join(_HERE, "lib", "libbvg_tools.so")
    src = os.path.join(_HERE, "bvg_store.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "lib/libbvg_tools.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i64, u64 = C.c_void_p, C.c_int64, C.c_uint64
        pp = C.POINTER(vp)
        L.bvgt_store.argtypes = [C.POINTER(Params), i64, vp, vp, i64, C.c_int, pp, C.POINTER(u64), pp, C.POINTER(Stats)]
        L.bvgt_synth_store.argtypes = [C.POINTER(Params), C.POINTER(SynthParams), i64, u64, i64, C.c_int, pp, C.POINTER(u64), pp, C.POINTER(Stats)]
        L.bvgt_synth_adjacency.argtypes = [C.POINTER(SynthParams), i64, u64, i64, pp, pp]
        L.bvgt_encode_offsets.argtypes = [vp, i64, C.c_int, pp, C.POINTER(u64), C.POINTER(u64)]
        L.bvgt_encode_values.argtypes = [vp, i64, C.c_int, C.c_int, pp, C.POINTER(u64)]
        L.bvgt_store_labels.argtypes = [C.c_int, C.c_int, vp, vp, i64, pp, C.POINTER(u64), pp]
        L.bvgt_store_label_lists.argtypes = [C.c_int, vp, vp, vp, i64, pp, C.POINTER(u64), pp]
        L.bvgt_store_label_lists64.argtypes = [C.c_int, vp, vp, vp, i64, pp, C.POINTER(u64), pp]
        L.bvgt_free.argtypes = [vp]
        _LIB = L
    return _LIB


def _take(ptr, nbytes, dtype=np.uint8):
    """Copies a malloc'ed buffer into numpy and frees it."""
    n = nbytes // np.dtype(dtype).itemsize
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,)).view(dtype)[:n].copy() if nbytes else np.empty(0, dtype)
    lib().bvgt_free(ptr)
    return arr


class Stored:
    """Result of a store: .graph bytes, offsets[n+1] (bit positions), params, stats."""

    def __init__(self, params, graph, offsets, stats):
        self.params, self.graph, self.offsets, self.stats = params, graph, offsets, stats

    def offsets_file(self):
        """The bytes BVGraph would write to basename.offsets (BVGraph.java:2228,2311)."""
        return encode_offsets(self.offsets, self.params.offset_coding)

    def properties_text(self):
        p = self.params
        names = {1: "DELTA", 2: "GAMMA", 3: "GOLOMB", 5: "UNARY", 6: "ZETA", 7: "NIBBLE"}
        d = default_params()
        flags = []
        for field, prefix in (("outdegree_coding", "OUTDEGREES_"), ("block_coding", "BLOCKS_"), ("residual_coding", "RESIDUALS_"),
                              ("reference_coding", "REFERENCES_"), ("block_count_coding", "BLOCK_COUNT_"), ("offset_coding", "OFFSETS_")):
            if getattr(p, field) != getattr(d, field):
                flags.append(prefix + names[getattr(p, field)])
        return ("#BVGraph properties\ngraphclass=it.unimi.dsi.big.webgraph.BVGraph\nversion=0\nnodes=%d\narcs=%d\n"
                "windowsize=%d\nmaxrefcount=%d\nminintervallength=%d\nzetak=%d\ncompressionflags=%s\n"
                % (p.nodes, p.arcs, p.window_size, p.max_ref_count, p.min_interval_length, p.zeta_k, " | ".join(flags)))

    def write(self, basename):
        with open(basename + ".graph", "wb") as f:
            f.write(self.graph.tobytes())
        with open(basename + ".offsets", "wb") as f:
            f.write(self.offsets_file().tobytes())
        with open(basename + ".properties", "w") as f:
            f.write(self.properties_text())


def store(adj_lists_or_csr, params=None, chunk_nodes=0, threads=1):
    """BVGraph.store over an adjacency: either a list of sorted lists or (adj_off uint64[n+1], adj int64[m])."""
    if isinstance(adj_lists_or_csr, tuple):
        off, adj = adj_lists_or_csr
        off = np.ascontiguousarray(off, dtype=np.uint64); adj = np.ascontiguousarray(adj, dtype=np.int64)
    else:
        lists = adj_lists_or_csr
        off = np.zeros(len(lists) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(l) for l in lists], dtype=np.uint64) if len(lists) else 0
        adj = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.int64) for l in lists]) if len(lists) and off[-1] else np.empty(0, np.int64), dtype=np.int64)
    n = len(off) - 1
    p = (params or default_params()).clone(nodes=n, arcs=int(off[-1]))
    g = C.c_void_p(); o = C.c_void_p(); gb = C.c_uint64(); st = Stats()
    adj_buf = adj if len(adj) else np.zeros(1, np.int64)
    r = lib().bvgt_store(C.byref(p), n, off.ctypes.data, adj_buf.ctypes.data, chunk_nodes, threads, C.byref(g), C.byref(gb), C.byref(o), C.byref(st))
    if r:
        raise RuntimeError("bvgt_store failed: %d" % r)
    return Stored(p, _take(g, gb.value), _take(o, 8 * (n + 1), np.uint64), st.as_dict())


def synth_store(n, seed=0, params=None, synth=None, chunk_nodes=1 << 16, threads=None):
    """Generates the synthetic web-like graph and stores it; returns Stored (arcs filled from the run)."""
    threads = threads or min(os.cpu_count() or 1, 64)
    p = (params or default_params()).clone(nodes=n)
    sp = synth or web_like()
    g = C.c_void_p(); o = C.c_void_p(); gb = C.c_uint64(); st = Stats()
    r = lib().bvgt_synth_store(C.byref(p), C.byref(sp), n, seed, chunk_nodes, threads, C.byref(g), C.byref(gb), C.byref(o), C.byref(st))
    if r:
        raise RuntimeError("bvgt_synth_store failed: %d" % r)
    p.arcs = st.arcs
    return Stored(p, _take(g, gb.value), _take(o, 8 * (n + 1), np.uint64), st.as_dict())


def tile_host(st, copies):
    """Host twin of bvg_tile: `copies` back-to-back copies of a stored graph as one Stored (BV records are translation invariant,
    SURVEY A.3: copy j is the base graph shifted by j * nodes).  Used to give the CPU baseline a stream of realistic size."""
    base = np.ascontiguousarray(st.graph, dtype=np.uint8)
    nbits = int(st.offsets[-1])
    total = nbits * copies
    out = np.zeros((total + 7) // 8 + 2, dtype=np.uint8)
    nb = (nbits + 7) // 8
    src = base[:nb].copy()
    if nbits & 7:
        src[-1] &= np.uint8((0xFF << (8 - (nbits & 7))) & 0xFF)          # padding bits of the last byte must stay clear
    for j in range(copies):
        o = j * nbits; b0 = o >> 3; sh = o & 7
        if sh == 0:
            out[b0:b0 + nb] |= src
        else:
            out[b0:b0 + nb] |= src >> np.uint8(sh)
            out[b0 + 1:b0 + 1 + nb] |= (src << np.uint8(8 - sh)) & np.uint8(0xFF)
    n = int(st.params.nodes)
    offs = np.empty(n * copies + 1, dtype=np.uint64)
    for j in range(copies):
        offs[j * n:(j + 1) * n] = st.offsets[:n] + np.uint64(j * nbits)
    offs[-1] = total
    p = st.params.clone(nodes=n * copies, arcs=int(st.stats["arcs"]) * copies)
    stats = dict(st.stats); stats["arcs"] = int(st.stats["arcs"]) * copies
    return Stored(p, out[:(total + 7) // 8], offs, stats)


def mosaic_host(sts, cycles):
    """Host twin of bvg_mosaic: the cycle of the stored graphs `sts`, back to back, repeated `cycles` times, as one Stored."""
    def put(out, o, src, nbits):
        nb = (nbits + 7) // 8
        piece = src[:nb].copy()
        if nbits & 7:
            piece[-1] &= np.uint8((0xFF << (8 - (nbits & 7))) & 0xFF)    # padding bits of the last byte must stay clear
        b0 = o >> 3; sh = o & 7
        if sh == 0:
            out[b0:b0 + nb] |= piece
        else:
            out[b0:b0 + nb] |= piece >> np.uint8(sh)
            out[b0 + 1:b0 + 1 + nb] |= (piece << np.uint8(8 - sh)) & np.uint8(0xFF)
    bits = [int(st.offsets[-1]) for st in sts]
    cyc_bits = sum(bits); total = cyc_bits * cycles
    out = np.zeros((total + 7) // 8 + 2, dtype=np.uint8)
    ns = [int(st.params.nodes) for st in sts]
    offs = np.empty(sum(ns) * cycles + 1, dtype=np.uint64)
    o = 0; i = 0
    for c in range(cycles):
        for st, nb, n in zip(sts, bits, ns):
            put(out, o, np.ascontiguousarray(st.graph, dtype=np.uint8), nb)
            offs[i:i + n] = st.offsets[:n] + np.uint64(o)
            o += nb; i += n
    offs[-1] = total
    arcs = sum(int(st.stats["arcs"]) for st in sts) * cycles
    p = sts[0].params.clone(nodes=sum(ns) * cycles, arcs=arcs)
    stats = dict(sts[0].stats); stats["arcs"] = arcs
    return Stored(p, out[:(total + 7) // 8], offs, stats)


def synth_adjacency(n, seed=0, synth=None, chunk_nodes=1 << 16):
    sp = synth or web_like()
    po = C.c_void_p(); pa = C.c_void_p()
    r = lib().bvgt_synth_adjacency(C.byref(sp), n, seed, chunk_nodes, C.byref(po), C.byref(pa))
    if r:
        raise RuntimeError("bvgt_synth_adjacency failed: %d" % r)
    off = _take(po, 8 * (n + 1), np.uint64)
    m = int(off[-1])
    adj = np.ctypeslib.as_array(C.cast(pa, C.POINTER(C.c_int64)), shape=(max(m, 1),))[:m].copy()
    lib().bvgt_free(pa)
    return off, adj


def encode_offsets(offsets, coding=GAMMA):
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    b = C.c_void_p(); nb = C.c_uint64(); nbits = C.c_uint64()
    r = lib().bvgt_encode_offsets(offsets.ctypes.data, len(offsets) - 1, coding, C.byref(b), C.byref(nb), C.byref(nbits))
    if r:
        raise RuntimeError("bvgt_encode_offsets failed: %d" % r)
    return _take(b, nb.value)


def encode_values(vals, coding, k=3):
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    b = C.c_void_p(); nb = C.c_uint64()
    r = lib().bvgt_encode_values(vals.ctypes.data if len(vals) else None, len(vals), coding, k, C.byref(b), C.byref(nb))
    if r:
        raise RuntimeError("bvgt_encode_values failed: %d" % r)
    return _take(b, nb.value)


class StoredLabels:
    """Result of store_labels: the .labels bytes, label_offsets[n+1] (bit positions) and the label class."""

    def __init__(self, kind, width, stream, offsets):
        self.kind, self.width, self.stream, self.offsets = kind, width, stream, offsets

    def spec(self):
        cls = "it.unimi.dsi.big.webgraph.labelling." + {1: "GammaCodedIntLabel", 2: "FixedWidthIntLabel", 3: "FixedWidthIntListLabel"}[self.kind]
        return cls + ("(FOO)" if self.kind == 1 else "(FOO,%d)" % self.width)

    def write(self, basename, underlying):
        """basename.{labels,labeloffsets,properties} as BitStreamArcLabelledImmutableGraph.store writes them (:655-700)."""
        with open(basename + ".labels", "wb") as f:
            f.write(self.stream.tobytes())
        with open(basename + ".labeloffsets", "wb") as f:
            f.write(encode_offsets(self.offsets, 2).tobytes())
        with open(basename + ".properties", "w") as f:
            f.write("graphclass = it.unimi.dsi.big.webgraph.labelling.BitStreamArcLabelledImmutableGraph\n"
                    "underlyinggraph = %s\nlabelspec = %s\n" % (underlying, self.spec()))


def store_labels(kind, width, values, arc_off):
    """Writes one int label per arc (values[m], in successor order; arc_off[n+1] = exclusive prefix of the outdegrees)."""
    values = np.ascontiguousarray(values, dtype=np.int32); arc_off = np.ascontiguousarray(arc_off, dtype=np.uint64)
    n = len(arc_off) - 1
    b = C.c_void_p(); o = C.c_void_p(); nb = C.c_uint64()
    vbuf = values if len(values) else np.zeros(1, np.int32)
    r = lib().bvgt_store_labels(kind, width, vbuf.ctypes.data, arc_off.ctypes.data, n, C.byref(b), C.byref(nb), C.byref(o))
    if r:
        raise RuntimeError("bvgt_store_labels failed: %d" % r)
    return StoredLabels(kind, width, _take(b, nb.value), _take(o, 8 * (n + 1), np.uint64))


def store_label_lists(width, list_off, values, arc_off):
    """Writes one int LIST per arc (FixedWidthIntListLabel): list_off[m+1] prefix of the list lengths, values the elements."""
    list_off = np.ascontiguousarray(list_off, dtype=np.uint64); values = np.ascontiguousarray(values, dtype=np.int32)
    arc_off = np.ascontiguousarray(arc_off, dtype=np.uint64)
    n = len(arc_off) - 1
    b = C.c_void_p(); o = C.c_void_p(); nb = C.c_uint64()
    vbuf = values if len(values) else np.zeros(1, np.int32)
    r = lib().bvgt_store_label_lists(width, list_off.ctypes.data, vbuf.ctypes.data, arc_off.ctypes.data, n, C.byref(b), C.byref(nb), C.byref(o))
    if r:
        raise RuntimeError("bvgt_store_label_lists failed: %d" % r)
    return StoredLabels(3, width, _take(b, nb.value), _take(o, 8 * (n + 1), np.uint64))


def store_label_long_lists(width, list_off, values, arc_off):
    """Writes one long LIST per arc (FixedWidthLongListLabel, width <= 64): list_off[m+1] prefix of the list lengths, values int64."""
    list_off = np.ascontiguousarray(list_off, dtype=np.uint64); values = np.ascontiguousarray(values, dtype=np.int64)
    arc_off = np.ascontiguousarray(arc_off, dtype=np.uint64)
    n = len(arc_off) - 1
    b = C.c_void_p(); o = C.c_void_p(); nb = C.c_uint64()
    vbuf = values if len(values) else np.zeros(1, np.int64)
    r = lib().bvgt_store_label_lists64(width, list_off.ctypes.data, vbuf.ctypes.data, arc_off.ctypes.data, n, C.byref(b), C.byref(nb), C.byref(o))
    if r:
        raise RuntimeError("bvgt_store_label_lists64 failed: %d" % r)
    return StoredLabels(4, width, _take(b, nb.value), _take(o, 8 * (n + 1), np.uint64))
