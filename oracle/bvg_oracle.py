"""ctypes loader for the CPU oracle (oracle/libbvg_oracle.so).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never from the product package.  See oracle/bvg_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Params(C.Structure):
    _fields_ = [("nodes", C.c_int64), ("arcs", C.c_int64), ("window_size", C.c_int32), ("max_ref_count", C.c_int32),
                ("min_interval_length", C.c_int32), ("zeta_k", C.c_int32), ("outdegree_coding", C.c_int32),
                ("block_coding", C.c_int32), ("residual_coding", C.c_int32), ("reference_coding", C.c_int32),
                ("block_count_coding", C.c_int32), ("offset_coding", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ScanResult(C.Structure):
    _fields_ = [("nodes", C.c_uint64), ("arcs", C.c_uint64), ("chk", C.c_uint64)]


class Bits(C.Structure):
    _fields_ = [("p", C.c_void_p), ("nbits", C.c_uint64), ("pos", C.c_uint64), ("err", C.c_int)]


def build(force=False):
    if os.environ.get("BVG_ORACLE_LIB"):                     # an alternative build of the same source (oracle/Makefile `asan`: tests/test_sanitizers.py)
        return os.environ["BVG_ORACLE_LIB"]
    so = os.path.join(_HERE, "libbvg_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("bvg_oracle.c", "bvg_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libbvg_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i64, u64 = C.c_void_p, C.c_int64, C.c_uint64
        L.bvgo_default_params.argtypes = [C.POINTER(Params)]
        L.bvgo_parse_properties.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Params)]
        L.bvgo_decode_offsets.argtypes = [vp, C.c_size_t, i64, C.c_int, vp]
        L.bvgo_load.argtypes = [C.c_char_p, C.POINTER(vp)]
        L.bvgo_open_mem.argtypes = [C.POINTER(Params), vp, u64, vp, C.POINTER(vp)]
        L.bvgo_close.argtypes = [vp]
        L.bvgo_info.argtypes = [vp, C.POINTER(Params)]
        L.bvgo_offsets.argtypes = [vp]; L.bvgo_offsets.restype = vp
        L.bvgo_graph_bytes.argtypes = [vp, C.POINTER(u64)]; L.bvgo_graph_bytes.restype = vp
        L.bvgo_outdegree.argtypes = [vp, i64]; L.bvgo_outdegree.restype = i64
        L.bvgo_successors.argtypes = [vp, i64, vp, i64]; L.bvgo_successors.restype = i64
        L.bvgo_node_iterator.argtypes = [vp, i64, C.POINTER(vp)]
        L.bvgo_iter_set_upper_bound.argtypes = [vp, i64]
        L.bvgo_iter_has_next.argtypes = [vp]
        L.bvgo_iter_next.argtypes = [vp]; L.bvgo_iter_next.restype = i64
        L.bvgo_iter_outdegree.argtypes = [vp]; L.bvgo_iter_outdegree.restype = i64
        L.bvgo_iter_successors.argtypes = [vp]; L.bvgo_iter_successors.restype = C.POINTER(i64)
        L.bvgo_iter_bit_position.argtypes = [vp]; L.bvgo_iter_bit_position.restype = u64
        L.bvgo_iter_free.argtypes = [vp]
        L.bvgo_decode_range.argtypes = [vp, i64, i64, vp, vp, u64, C.POINTER(u64)]
        L.bvgo_mix.argtypes = [u64, u64]; L.bvgo_mix.restype = u64
        L.bvgo_scan.argtypes = [vp, i64, i64, u64, C.POINTER(ScanResult)]
        L.bvgo_scan_mt.argtypes = [vp, i64, i64, u64, C.c_int, C.POINTER(ScanResult)]
        L.bvgo_strerror.argtypes = [C.c_int]; L.bvgo_strerror.restype = C.c_char_p
        L.bvgo_bits_init.argtypes = [C.POINTER(Bits), vp, u64, u64]
        for name in ("unary", "gamma", "delta", "nibble"):
            f = getattr(L, "bvgo_read_" + name); f.argtypes = [C.POINTER(Bits)]; f.restype = u64
        L.bvgo_read_bits.argtypes = [C.POINTER(Bits), C.c_int]; L.bvgo_read_bits.restype = u64
        L.bvgo_read_zeta.argtypes = [C.POINTER(Bits), C.c_int]; L.bvgo_read_zeta.restype = u64
        L.bvgo_read_golomb.argtypes = [C.POINTER(Bits), u64]; L.bvgo_read_golomb.restype = u64
        L.bvgo_nat2int.argtypes = [u64]; L.bvgo_nat2int.restype = i64
        L.bvgo_parse_label_spec.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.bvgo_labels_decode.argtypes = [C.c_int, C.c_int, vp, u64, vp, i64, i64, i64, vp, vp, u64, C.POINTER(u64)]
        L.bvgo_labels_decode_lists.argtypes = [C.c_int, vp, u64, vp, i64, i64, i64, vp, vp, vp, u64, C.POINTER(u64)]
        L.bvgo_labels_decode_lists64.argtypes = [C.c_int, vp, u64, vp, i64, i64, i64, vp, vp, vp, u64, C.POINTER(u64)]
        _LIB = L
    return _LIB


class OracleError(Exception):
    def __init__(self, code):
        self.code = code
        super().__init__("oracle error %d: %s" % (code, lib().bvgo_strerror(code).decode()))


def _chk(r):
    if r < 0:
        raise OracleError(int(r))
    return r


def parse_properties(text):
    if isinstance(text, str):
        text = text.encode()
    p = Params()
    _chk(lib().bvgo_parse_properties(text, len(text), C.byref(p)))
    return p


def decode_offsets(obytes, nodes, coding=2):
    buf = np.frombuffer(bytes(obytes) + b"\0" * 16, dtype=np.uint8)
    out = np.empty(nodes + 1, dtype=np.uint64)
    _chk(lib().bvgo_decode_offsets(buf.ctypes.data, len(obytes), nodes, coding, out.ctypes.data))
    return out


class Graph:
    """Oracle-side BVGraph (mirrors ImmutableGraph.load / BVGraph API on the CPU)."""

    def __init__(self, handle, keep=()):
        self._h = handle
        self._keep = keep

    @classmethod
    def load(cls, basename):
        h = C.c_void_p()
        _chk(lib().bvgo_load(os.fsencode(basename), C.byref(h)))
        return cls(h)

    @classmethod
    def from_memory(cls, params, graph_bytes, offsets=None):
        g = np.frombuffer(bytes(graph_bytes) + b"\0" * 16, dtype=np.uint8)
        o = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        _chk(lib().bvgo_open_mem(C.byref(params), g.ctypes.data, len(graph_bytes), None if o is None else o.ctypes.data, C.byref(h)))
        return cls(h, keep=(g, o, params))

    def close(self):
        if self._h:
            lib().bvgo_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def params(self):
        p = Params(); lib().bvgo_info(self._h, C.byref(p)); return p

    def num_nodes(self):
        return self.params.nodes

    def outdegree(self, x):
        return int(_chk(lib().bvgo_outdegree(self._h, x)))

    def successors(self, x):
        d = self.outdegree(x)
        out = np.empty(max(d, 1), dtype=np.int64)
        _chk(lib().bvgo_successors(self._h, x, out.ctypes.data, d))
        return out[:d]

    def offsets(self):
        n = self.params.nodes
        p = lib().bvgo_offsets(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()

    def graph_bytes(self):
        n = C.c_uint64()
        p = lib().bvgo_graph_bytes(self._h, C.byref(n))
        return C.string_at(p, n.value)

    def decode_range(self, frm, to, total_hint=None):
        """Returns (outdeg int32[to-frm], succ int64[sum d]) via the sequential iterator."""
        outd = np.empty(max(to - frm, 1), dtype=np.int32)
        n = C.c_uint64()
        _chk(lib().bvgo_decode_range(self._h, frm, to, outd.ctypes.data, None, 0, C.byref(n)))
        succ = np.empty(max(n.value, 1), dtype=np.int64)
        _chk(lib().bvgo_decode_range(self._h, frm, to, outd.ctypes.data, succ.ctypes.data, n.value, C.byref(n)))
        return outd[:to - frm], succ[:n.value]

    def scan(self, frm=0, to=None, node_base=0, threads=1):
        to = self.params.nodes if to is None else to
        r = ScanResult()
        if threads <= 1:
            _chk(lib().bvgo_scan(self._h, frm, to, node_base, C.byref(r)))
        else:
            _chk(lib().bvgo_scan_mt(self._h, frm, to, node_base, threads, C.byref(r)))
        return {"nodes": r.nodes, "arcs": r.arcs, "chk": r.chk}

    def node_iterator(self, frm=0, upper=None):
        return NodeIterator(self, frm, upper)


class NodeIterator:
    def __init__(self, g, frm, upper=None):
        self._g = g
        self._h = C.c_void_p()
        _chk(lib().bvgo_node_iterator(g._h, frm, C.byref(self._h)))
        if upper is not None:
            lib().bvgo_iter_set_upper_bound(self._h, upper)

    def has_next(self):
        return bool(lib().bvgo_iter_has_next(self._h))

    def next(self):
        x = lib().bvgo_iter_next(self._h)
        if x == -1:
            raise StopIteration
        return int(_chk(x))

    def outdegree(self):
        return int(_chk(lib().bvgo_iter_outdegree(self._h)))

    def successors(self):
        d = self.outdegree()
        p = lib().bvgo_iter_successors(self._h)
        return np.ctypeslib.as_array(p, shape=(max(d, 1),))[:d].copy()

    def bit_position(self):
        return int(lib().bvgo_iter_bit_position(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().bvgo_iter_free(self._h); self._h = None
        except Exception:
            pass


def mix(x, y):
    return int(lib().bvgo_mix(x, y))


LABEL_GAMMA_INT, LABEL_FIXED_INT, LABEL_FIXED_INT_LIST = 1, 2, 3


def parse_label_spec(spec):
    """Label.toSpec() text -> (kind, width)."""
    k, w = C.c_int(), C.c_int()
    _chk(lib().bvgo_parse_label_spec(spec.encode() if isinstance(spec, str) else spec, C.byref(k), C.byref(w)))
    return k.value, w.value


def labels_decode(kind, width, stream, loffsets, frm, to, outdeg):
    """Labels of the arcs of nodes [frm,to) (BitStreamArcLabelledImmutableGraph.java:565-582): int32 array."""
    stream = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
    pad = np.concatenate([stream, np.zeros(16, np.uint8)])
    lo = np.ascontiguousarray(loffsets, dtype=np.uint64)
    deg = np.ascontiguousarray(outdeg, dtype=np.int32)
    total = int(deg.sum())
    out = np.empty(max(total, 1), dtype=np.int32)
    n = C.c_uint64()
    _chk(lib().bvgo_labels_decode(kind, width, pad.ctypes.data, len(stream), lo.ctypes.data, len(lo) - 1, frm, to,
                                  deg.ctypes.data if len(deg) else None, out.ctypes.data, total, C.byref(n)))
    return out[:total]


def labels_decode_lists(width, stream, loffsets, frm, to, outdeg):
    """List labels (FixedWidthIntListLabel.java:73-78) of the arcs of nodes [frm,to): (list_off uint64[arcs+1], values int32)."""
    stream = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
    pad = np.concatenate([stream, np.zeros(16, np.uint8)])
    lo = np.ascontiguousarray(loffsets, dtype=np.uint64)
    deg = np.ascontiguousarray(outdeg, dtype=np.int32)
    arcs = int(deg.sum())
    loff = np.zeros(arcs + 1, dtype=np.uint64)
    n = C.c_uint64()
    args = (width, pad.ctypes.data, len(stream), lo.ctypes.data, len(lo) - 1, frm, to, deg.ctypes.data if len(deg) else None, loff.ctypes.data)
    _chk(lib().bvgo_labels_decode_lists(*args, None, 0, C.byref(n)))
    vals = np.empty(max(n.value, 1), dtype=np.int32)
    _chk(lib().bvgo_labels_decode_lists(*args, vals.ctypes.data, n.value, C.byref(n)))
    return loff, vals[:n.value]


def labels_decode_long_lists(width, stream, loffsets, frm, to, outdeg):
    """Long list labels (FixedWidthLongListLabel.java:81-87) of the arcs of nodes [frm,to): (list_off uint64[arcs+1], values int64)."""
    stream = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
    pad = np.concatenate([stream, np.zeros(16, np.uint8)])
    lo = np.ascontiguousarray(loffsets, dtype=np.uint64)
    deg = np.ascontiguousarray(outdeg, dtype=np.int32)
    arcs = int(deg.sum())
    loff = np.zeros(arcs + 1, dtype=np.uint64)
    n = C.c_uint64()
    args = (width, pad.ctypes.data, len(stream), lo.ctypes.data, len(lo) - 1, frm, to, deg.ctypes.data if len(deg) else None, loff.ctypes.data)
    _chk(lib().bvgo_labels_decode_lists64(*args, None, 0, C.byref(n)))
    vals = np.empty(max(n.value, 1), dtype=np.int64)
    _chk(lib().bvgo_labels_decode_lists64(*args, vals.ctypes.data, n.value, C.byref(n)))
    return loff, vals[:n.value]
