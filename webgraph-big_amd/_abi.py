"""ctypes mirror of include/bvgraph_hip.h (struct layouts + status codes)."""
import ctypes as C


class Params(C.Structure):
    """bvg_params (include/bvgraph_hip.h)."""
    _fields_ = [("nodes", C.c_int64), ("arcs", C.c_int64), ("window_size", C.c_int32), ("max_ref_count", C.c_int32),
                ("min_interval_length", C.c_int32), ("zeta_k", C.c_int32), ("outdegree_coding", C.c_int32),
                ("block_coding", C.c_int32), ("residual_coding", C.c_int32), ("reference_coding", C.c_int32),
                ("block_count_coding", C.c_int32), ("offset_coding", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    def clone(self, **kw):
        p = Params()
        C.memmove(C.byref(p), C.byref(self), C.sizeof(Params))
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class ScanResult(C.Structure):
    """bvg_scan_result."""
    _fields_ = [("nodes", C.c_uint64), ("arcs", C.c_uint64), ("chk", C.c_uint64), ("graph_bytes", C.c_uint64),
                ("index_bytes", C.c_uint64), ("kernel_ms", C.c_double), ("launches", C.c_uint32), ("slow_blocks", C.c_uint32),
                ("index_entries", C.c_uint64), ("lean_blocks", C.c_uint32), ("reserved0", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Tuning(C.Structure):
    """bvg_tuning."""
    _fields_ = [("block_bits", C.c_uint32), ("force_wide", C.c_uint32), ("force_slow", C.c_uint32), ("reserved", C.c_uint32), ("no_index", C.c_uint32)]


DELTA, GAMMA, GOLOMB, SKEWED_GOLOMB, UNARY, ZETA, NIBBLE = 1, 2, 3, 4, 5, 6, 7
LOAD_OFFLINE, LOAD_SEQUENTIAL, LOAD_STANDARD, LOAD_MAPPED = -1, 0, 1, 2

OK, E_ARG, E_STATE, E_UNSUPPORTED, E_IO, E_EOF, E_NOMEM, E_HIP, E_CAPACITY = 0, -1, -2, -3, -4, -5, -6, -7, -8


def default_params(**kw):
    """BVGraph defaults (BVGraph.java:455-473, 527-542)."""
    p = Params(nodes=0, arcs=-1, window_size=7, max_ref_count=3, min_interval_length=4, zeta_k=3,
               outdegree_coding=GAMMA, block_coding=GAMMA, residual_coding=ZETA, reference_coding=UNARY,
               block_count_coding=GAMMA, offset_coding=GAMMA)
    for k, v in kw.items():
        setattr(p, k, v)
    return p
