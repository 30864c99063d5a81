"""Host-side mirror of the reference's graph API over the C ABI of libbvgraph_hip.so.

Class and method names follow the reference (paths relative to /root/reference/src/it/unimi/dsi/big/webgraph):
  ImmutableGraph.java:245-447   numNodes / numArcs / randomAccess / outdegree / successors /
                                successorBigArray / nodeIterator / splitNodeIterators / copy
  NodeIterator.java:34-133      hasNext / nextLong / outdegree / successors / successorBigArray / copy(upperBound) / skip
  LazyLongIterator.java:28-44   nextLong() returns -1 at the end; skip(n)
  BVGraph.java:1345-1464        load / loadMapped / loadOffline / loadSequential

Every decode goes through the HIP kernels (bvg_decode_range / bvg_scan); there is no CPU decode in
this package and importing it on a box without the built library or without a GPU fails loudly at
the first call that needs the device.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

from . import _abi
from ._abi import Params, ScanResult, Tuning

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class BVGraphError(Exception):
    """Base of the status-code exceptions; subclasses mirror the Java exception classes."""

    def __init__(self, status, what=""):
        self.status = status
        msg = lib().bvg_strerror(status).decode() if _LIB is not None else str(status)
        super().__init__("%s [%d] %s" % (what, status, msg))


class IllegalArgumentException(BVGraphError, ValueError):
    pass


class IllegalStateException(BVGraphError, RuntimeError):
    pass


class UnsupportedOperationException(BVGraphError, NotImplementedError):
    pass


class IOException(BVGraphError, OSError):
    pass


class EOFException(IOException):
    pass


class DeviceError(BVGraphError):
    pass


class NoSuchElementException(StopIteration):
    pass


_EXC = {_abi.E_ARG: IllegalArgumentException, _abi.E_STATE: IllegalStateException, _abi.E_UNSUPPORTED: UnsupportedOperationException,
        _abi.E_IO: IOException, _abi.E_EOF: EOFException, _abi.E_NOMEM: MemoryError, _abi.E_HIP: DeviceError}


def _check(status, what=""):
    if status == 0:
        return
    exc = _EXC.get(status, BVGraphError)
    if exc is MemoryError:
        raise MemoryError("%s: out of host/device memory" % what)
    raise exc(status, what)


def library_path():
    # BVG_HIP_LIB: an alternative build of the same library (e.g. lib/libbvgraph_hip_prof.so, `make prof`: cycle timers compiled in)
    return os.environ.get("BVG_HIP_LIB") or os.path.join(_HERE, "lib", "libbvgraph_hip.so")


def build(force=False):
    """Compiles libbvgraph_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    so = library_path()
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))] + [os.path.join(_HERE, "..", "include", "bvgraph_hip.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "lib/libbvgraph_hip.so"])
    return so


def lib():
    """Loads the HIP library; raises if it is missing (no fallback)."""
    global _LIB
    if _LIB is None:
        so = library_path()
        if not os.path.exists(so):
            raise ImportError("libbvgraph_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        L = C.CDLL(so)
        vp, i64, u64, pp = C.c_void_p, C.c_int64, C.c_uint64, C.POINTER(C.c_void_p)
        L.bvg_abi_version.restype = C.c_int
        L.bvg_default_params.argtypes = [C.POINTER(Params)]
        L.bvg_parse_properties.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Params)]
        L.bvg_decode_offsets.argtypes = [vp, C.c_size_t, i64, C.c_int, vp]
        L.bvg_open.argtypes = [C.c_char_p, C.c_int, C.c_int, pp]
        L.bvg_open_mem.argtypes = [C.POINTER(Params), vp, u64, vp, C.c_int, pp]
        L.bvg_open_dev.argtypes = [C.POINTER(Params), vp, u64, vp, C.c_int, pp]
        L.bvg_copy.argtypes = [vp, pp]
        L.bvg_close.argtypes = [vp]; L.bvg_close.restype = None
        L.bvg_info.argtypes = [vp, C.POINTER(Params)]
        L.bvg_set_node_base.argtypes = [vp, u64]
        L.bvg_get_offsets.argtypes = [vp, vp]
        L.bvg_outdegrees.argtypes = [vp, i64, i64, vp]
        L.bvg_decode_range.argtypes = [vp, i64, i64, vp, vp, u64, C.POINTER(u64)]
        L.bvg_decode_range_dev.argtypes = [vp, i64, i64, vp, vp, u64, C.POINTER(u64)]
        L.bvg_decode_range32.argtypes = [vp, i64, i64, vp, vp, u64, C.POINTER(u64)]
        L.bvg_scan.argtypes = [vp, i64, i64, C.POINTER(ScanResult)]
        L.bvg_successors_batch.argtypes = [vp, vp, i64, vp, vp, u64, C.POINTER(u64)]
        L.bvg_host_alloc.argtypes = [C.c_size_t]; L.bvg_host_alloc.restype = vp
        L.bvg_store.argtypes = [C.POINTER(Params), i64, vp, vp, i64, C.c_int, pp, C.POINTER(u64), pp]
        L.bvg_free.argtypes = [vp]; L.bvg_free.restype = None
        L.bvg_host_free.argtypes = [vp]; L.bvg_host_free.restype = None
        L.bvg_split_by_bits.argtypes = [vp, C.c_int, vp]
        L.bvg_split_by_arcs.argtypes = [vp, C.c_int, vp]
        L.bvg_shard_bounds.argtypes = [vp, C.c_int, C.c_int, vp]
        L.bvg_scan_shard.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(ScanResult), C.POINTER(i64), C.POINTER(i64)]
        L.bvg_scan_multi.argtypes = [vp, C.c_int, C.c_int, C.POINTER(ScanResult), vp]
        L.bvg_transpose.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
        L.bvg_transpose_dev.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
        L.bvg_symmetrize.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
        L.bvg_symmetrize_dev.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
        L.bvg_labels_parse_spec.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.bvg_labels_read_properties.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.bvg_labels_open_mem.argtypes = [C.c_int, C.c_int, i64, vp, u64, vp, C.c_int, pp]
        L.bvg_labels_open.argtypes = [C.c_char_p, i64, C.c_int, pp, C.c_char_p, C.c_size_t]
        L.bvg_labels_close.argtypes = [vp]; L.bvg_labels_close.restype = None
        L.bvg_labels_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(u64)]
        L.bvg_labels_decode_range.argtypes = [vp, i64, i64, vp, vp, u64, C.POINTER(u64)]
        L.bvg_labels_decode_range_dev.argtypes = [vp, i64, i64, vp, vp, u64, C.POINTER(u64)]
        L.bvg_labels_decode_range_lists.argtypes = [vp, i64, i64, vp, vp, vp, u64, C.POINTER(u64)]
        L.bvg_labels_decode_range_lists64.argtypes = [vp, i64, i64, vp, vp, vp, u64, C.POINTER(u64)]
        L.bvg_tile.argtypes = [vp, i64, pp]
        L.bvg_mosaic.argtypes = [vp, C.c_int, i64, pp]
        L.bvg_set_tuning.argtypes = [vp, C.POINTER(Tuning)]
        L.bvg_strerror.argtypes = [C.c_int]; L.bvg_strerror.restype = C.c_char_p
        L.bvg_arc_mix.argtypes = [u64, u64]; L.bvg_arc_mix.restype = u64
        L.bvg_build_index.argtypes = [vp, i64, i64, C.POINTER(u64), C.POINTER(u64)]
        L.bvg_save_index.argtypes = [vp, C.c_char_p]
        L.bvg_load_index.argtypes = [vp, C.c_char_p]
        if L.bvg_abi_version() != 3:
            raise ImportError("libbvgraph_hip.so ABI mismatch")
        _LIB = L
    return _LIB


def store(adj, params=None, chunk_nodes=0, device=0):
    """BVGraph.store on the device (bvg_store): adj = (adj_off uint64[n+1], succ int64[m]) or a list of sorted lists.
    Returns (graph uint8[], offsets uint64[n+1]); byte for byte what the reference's compressor writes."""
    if isinstance(adj, tuple):
        off = np.ascontiguousarray(adj[0], dtype=np.uint64); succ = np.ascontiguousarray(adj[1], dtype=np.int64)
    else:
        off = np.zeros(len(adj) + 1, dtype=np.uint64)
        if len(adj):
            off[1:] = np.cumsum([len(l) for l in adj], dtype=np.uint64)
        succ = np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.int64) for l in adj]) if len(adj) and off[-1] else np.empty(0, np.int64), dtype=np.int64)
    n = len(off) - 1
    p = params if params is not None else _abi.default_params()
    g = C.c_void_p(); o = C.c_void_p(); nb = C.c_uint64()
    sb = succ if len(succ) else np.zeros(1, np.int64)
    _check(lib().bvg_store(C.byref(p), n, off.ctypes.data, sb.ctypes.data, chunk_nodes, device, C.byref(g), C.byref(nb), C.byref(o)), "store")
    try:
        graph = np.ctypeslib.as_array(C.cast(g, C.POINTER(C.c_uint8)), shape=(max(int(nb.value), 1),))[:int(nb.value)].copy()
        offsets = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
    finally:
        lib().bvg_free(g); lib().bvg_free(o)
    return graph, offsets


def parse_properties(text):
    if isinstance(text, str):
        text = text.encode()
    p = Params()
    _check(lib().bvg_parse_properties(text, len(text), C.byref(p)), "parse_properties")
    return p


def decode_offsets(obytes, nodes, coding=_abi.GAMMA):
    buf = np.frombuffer(bytes(obytes), dtype=np.uint8)
    out = np.empty(nodes + 1, dtype=np.uint64)
    _check(lib().bvg_decode_offsets(buf.ctypes.data if len(buf) else None, len(buf), nodes, coding, out.ctypes.data), "decode_offsets")
    return out


def arc_mix(x, y):
    return int(lib().bvg_arc_mix(x, y))


class LazyLongIterator:
    """LazyLongIterator.java:28-44 over a decoded successor array (LazyLongIterators.wrap, :220-255)."""

    def __init__(self, arr):
        self._a = arr
        self._i = 0

    def next_long(self):
        if self._i >= len(self._a):
            return -1
        v = int(self._a[self._i]); self._i += 1
        return v

    nextLong = next_long

    def skip(self, n):
        k = min(int(n), len(self._a) - self._i)
        self._i += k
        return k

    def __iter__(self):
        while True:
            v = self.next_long()
            if v == -1:
                return
            yield v


class PinnedArray:
    """A numpy view over page-locked host memory (bvg_host_alloc): device -> host copies into it run at the PCIe rate."""

    def __init__(self, count, dtype):
        self.dtype = np.dtype(dtype)
        self.nbytes = max(int(count), 1) * self.dtype.itemsize
        self._p = lib().bvg_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError("bvg_host_alloc(%d)" % self.nbytes)
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(self.nbytes,)).view(self.dtype)

    def close(self):
        if getattr(self, "_p", None):
            self.array = None
            lib().bvg_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_PIN_CACHE = {}          # nbytes -> [PinnedArray]: page-locking memory costs ~0.3 s per GiB, so iterators hand their buffers on
_PIN_CACHE_LIMIT = 8 << 30
_pin_cached = 0
_PIN_LOCK = threading.Lock()   # iterators recycle buffers from their helper threads too


def _pinned(count, dtype):
    global _pin_cached
    nbytes = max(int(count), 1) * np.dtype(dtype).itemsize
    with _PIN_LOCK:
        lst = _PIN_CACHE.get(nbytes)
        pa = lst.pop() if lst else None
        if pa is not None:
            _pin_cached -= nbytes
    if pa is not None:
        pa.dtype = np.dtype(dtype)
        pa.array = np.ctypeslib.as_array(C.cast(pa._p, C.POINTER(C.c_uint8)), shape=(nbytes,)).view(pa.dtype)
        return pa
    return PinnedArray(count, dtype)


def _unpin(pa):
    """Hands a page-locked buffer back.  Views of it that a caller kept (successor_array(), batch()) are valid only until the
    iterator moves on (NodeIterator.java:80-96: "the returned array may be reused"): copy what must outlive that."""
    global _pin_cached
    if pa is None or not getattr(pa, "_p", None):
        return
    with _PIN_LOCK:
        keep = _pin_cached + pa.nbytes <= _PIN_CACHE_LIMIT
        if keep:
            _PIN_CACHE.setdefault(pa.nbytes, []).append(pa); _pin_cached += pa.nbytes
    if not keep:
        pa.close()


class _BatchSlot:
    """One of the two batch buffers of a NodeIterator: a flyweight handle of its own (so the decode of the next batch can run on
    another host thread, ImmutableGraph.java:187-197) and page-locked outdegree / successor buffers that are reused."""

    def __init__(self, graph, batch_nodes):
        self.g = graph.copy()
        self.deg = _pinned(batch_nodes, np.int32)
        # graphs whose ids fit 32 bits cross PCIe as uint32 (bvg_decode_range32): the transfer bounds this path, so half the bytes is
        # twice the rate; successor_array() widens to the longs of NodeIterator.successorBigArray()
        self.dt = np.uint32 if graph.num_nodes() + graph.node_base() <= 0xFFFFFFFF else np.int64   # (0xFFFFFFFF is never an id on this transport: it stands for -1)
        self.succ = _pinned(max(1024, 32 * batch_nodes), self.dt)
        self.lo = self.hi = 0
        self.n_succ = 0

    def decode(self, lo, hi):
        need = C.c_uint64(0)
        fn = lib().bvg_decode_range32 if self.dt is np.uint32 else lib().bvg_decode_range
        while True:
            st = fn(self.g._h, lo, hi, self.deg.array.ctypes.data, self.succ.array.ctypes.data, len(self.succ.array), C.byref(need))
            if st == _abi.E_CAPACITY:
                _unpin(self.succ)
                self.succ = _pinned(((int(need.value) + int(need.value) // 4) + 0xFFFFF) & ~0xFFFFF, self.dt)
                continue
            _check(st, "decode_range(%d,%d)" % (lo, hi))
            break
        self.lo, self.hi, self.n_succ = lo, hi, int(need.value)
        return self

    def close(self):
        _unpin(self.deg); _unpin(self.succ); self.deg = self.succ = None; self.g.close()


class NodeIterator:
    """BVGraph.BVGraphNodeIterator (BVGraph.java:1100-1245) fed by batched GPU decodes.  Two batch slots: while the caller walks
    batch i, a helper thread decodes batch i+1 (kernels + device -> host copy into page-locked memory) through a flyweight of
    the graph, so the PCIe transfer and the decode overlap with the consumer."""

    def __init__(self, graph, frm, upper_bound=None, batch_nodes=None):
        n = graph.num_nodes()
        if frm < 0 or frm > n:
            raise IllegalArgumentException(_abi.E_ARG, "nodeIterator(%d)" % frm)        # BVG:1128
        self._g = graph
        self._from = frm
        self._curr = frm - 1                                                           # BVG:1147
        self._limit = min(upper_bound if upper_bound is not None else n, n) - 1         # BVG:1148
        self._batch_nodes = batch_nodes or graph.iterator_batch_nodes
        self._b0 = frm; self._b1 = frm
        self._deg = None; self._cum = None; self._succ = None
        self._slots = None; self._pending = []; self._pool = None; self._held = None; self._free = []

    def has_next(self):
        return self._curr < self._limit                                                # BVG:1179-1181

    hasNext = has_next

    _DEPTH = int(os.environ.get("BVG_ITER_DEPTH", "2"))                                                                         # batches decoded ahead of the caller

    def _start(self):
        import concurrent.futures
        bn = max(1, min(self._batch_nodes, self._limit + 1 - self._from))
        nb = -(-(self._limit + 1 - self._from) // bn)
        depth = max(0, min(self._DEPTH, nb - 1))
        self._slots = [_BatchSlot(self._g, bn) for _ in range(depth + 1)]
        self._free = list(range(depth + 1))
        self._pending = []                                                             # [(future, first node, slot index)], in node order
        self._held = None
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, depth)) if depth else None

    def _fill(self, x):
        if self._slots is None:
            self._start()
        if self._held is not None:                                                     # the batch the caller has left: its buffers may be reused
            self._free.append(self._held); self._held = None
        slot = None
        while self._pending:
            fut, plo, si = self._pending.pop(0)
            got = fut.result()                                                         # (raises what the decode raised)
            if plo == x:
                slot = got; self._held = si
                break
            self._free.append(si)                                                      # the caller jumped elsewhere: drop what was decoded ahead
        if slot is None:
            si = self._free.pop()
            slot = self._slots[si].decode(x, min(x + self._batch_nodes, self._limit + 1)); self._held = si
        # keep the pipeline full: the batches behind this one are decoded (kernels + device -> host copy) by helper threads, each
        # through its own flyweight handle and stream, so one batch's copy overlaps the next one's kernels
        nxt = self._pending[-1][1] + self._batch_nodes if self._pending else slot.hi
        while self._pool is not None and self._free and nxt <= self._limit:
            si = self._free.pop()
            self._pending.append((self._pool.submit(self._slots[si].decode, nxt, min(nxt + self._batch_nodes, self._limit + 1)), nxt, si))
            nxt += self._batch_nodes
        self._b0, self._b1 = slot.lo, slot.hi
        self._deg = slot.deg.array[:slot.hi - slot.lo]
        self._cum = np.zeros(len(self._deg) + 1, dtype=np.int64)
        np.cumsum(self._deg, out=self._cum[1:])
        self._succ = slot.succ.array[:slot.n_succ]

    def next_long(self):
        if not self.has_next():
            raise NoSuchElementException()                                             # BVG:1165
        self._curr += 1
        if not (self._b0 <= self._curr < self._b1):
            self._fill(self._curr)
        return self._curr

    nextLong = next_long

    def __iter__(self):
        while self.has_next():
            yield self.next_long()

    def _require_started(self):
        if self._curr == self._from - 1:
            raise IllegalStateException(_abi.E_STATE, "no node fetched yet")            # BVG:1185,1193,1207

    def outdegree(self):
        self._require_started()
        return int(self._deg[self._curr - self._b0])

    def successor_array(self):
        """successorBigArray(): view valid until the next next_long() (NodeIterator.java:80-96)."""
        self._require_started()
        i = self._curr - self._b0
        v = self._succ[self._cum[i]:self._cum[i + 1]]
        if v.dtype == np.int64:
            return v
        w = v.astype(np.int64)                                         # (ids that crossed PCIe as uint32 are widened here)
        w[v == 0xFFFFFFFF] = -1                                        # the stand-in for a missing successor of a malformed stream: -1, as on the int64 transport
        return w

    successorBigArray = successor_array

    def batch(self):
        """The whole current batch at once: (first node, outdeg int32[], cum int64[], succ) — views valid until the next next_long()
        that leaves the batch (what a bulk consumer walks instead of one node at a time).  succ is uint32 for graphs whose ids fit
        32 bits (as it crossed PCIe: 0xFFFFFFFF there stands for the -1 of a malformed stream, never for a node), int64 otherwise."""
        self._require_started()
        return self._b0, self._deg, self._cum, self._succ

    def skip_batch(self):
        """Moves to the last node of the current batch (the next next_long() fetches the following batch)."""
        self._require_started()
        self._curr = self._b1 - 1

    def successors(self):
        return LazyLongIterator(self.successor_array())

    def copy(self, upper_bound=None):
        """NodeIterator.copy(upperBound), BVG:1223-1229: a new iterator positioned after the current node."""
        ub = self._limit + 1 if upper_bound is None else upper_bound
        return NodeIterator(self._g.copy(), self._curr + 1, ub, self._batch_nodes)

    def skip(self, n):
        k = 0
        while k < n and self.has_next():
            self.next_long(); k += 1
        return k

    def close(self):
        for fut, _, _ in self._pending:
            try:
                fut.result()
            except Exception:
                pass
        self._pending = []
        if self._pool is not None:
            self._pool.shutdown(wait=True); self._pool = None
        if self._slots is not None:
            for sl in self._slots:
                sl.close()
            self._slots = None
        self._deg = self._cum = self._succ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BVGraph:
    """ImmutableGraph / BVGraph surface for the decode path, backed by HBM-resident data."""

    iterator_batch_nodes = 1 << 16

    def __init__(self, handle, keep=()):
        self._h = handle
        self._keep = keep
        self._params = Params()
        _check(lib().bvg_info(self._h, C.byref(self._params)), "info")
        self._basename = None

    # ---- loading (BVGraph.java:1345-1464) ----
    @classmethod
    def load(cls, basename, device=0, mode=_abi.LOAD_STANDARD):
        h = C.c_void_p()
        _check(lib().bvg_open(os.fsencode(basename), mode, device, C.byref(h)), "load(%s)" % basename)
        g = cls(h); g._basename = basename
        return g

    @classmethod
    def load_mapped(cls, basename, device=0):
        return cls.load(basename, device, _abi.LOAD_MAPPED)

    @classmethod
    def load_offline(cls, basename, device=0):
        return cls.load(basename, device, _abi.LOAD_OFFLINE)

    @classmethod
    def load_sequential(cls, basename, device=0):
        return cls.load(basename, device, _abi.LOAD_SEQUENTIAL)

    @classmethod
    def from_memory(cls, params, graph_bytes, offsets, device=0):
        g = np.frombuffer(bytes(graph_bytes), dtype=np.uint8) if not isinstance(graph_bytes, np.ndarray) else np.ascontiguousarray(graph_bytes, dtype=np.uint8)
        o = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().bvg_open_mem(C.byref(params), g.ctypes.data if len(g) else None, len(g), None if o is None else o.ctypes.data, device, C.byref(h)), "open_mem")
        return cls(h)

    @classmethod
    def from_device(cls, params, d_graph_ptr, nbytes, d_offsets_ptr, device=0, keep=()):
        """Adopts buffers already resident in HBM (e.g. torch tensors; pass them in `keep`)."""
        h = C.c_void_p()
        _check(lib().bvg_open_dev(C.byref(params), d_graph_ptr, nbytes, d_offsets_ptr, device, C.byref(h)), "open_dev")
        return cls(h, keep=keep)

    def tile(self, copies):
        """Synthetic workload helper: `copies` back-to-back copies of this graph (bvg_tile)."""
        h = C.c_void_p()
        _check(lib().bvg_tile(self._h, copies, C.byref(h)), "tile")
        return BVGraph(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().bvg_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ImmutableGraph surface ----
    @property
    def params(self):
        return self._params

    def num_nodes(self):
        return int(self._params.nodes)

    def num_arcs(self):
        if self._params.arcs < 0:
            raise UnsupportedOperationException(_abi.E_UNSUPPORTED, "numArcs")           # ImmutableGraph.java:253-258
        return int(self._params.arcs)

    numNodes, numArcs = num_nodes, num_arcs

    def random_access(self):
        return True

    def has_copiable_iterators(self):
        return True

    def basename(self):
        return self._basename

    def window_size(self):
        return int(self._params.window_size)

    def max_ref_count(self):
        return int(self._params.max_ref_count)

    def min_interval_length(self):
        return int(self._params.min_interval_length)

    def copy(self):
        """BVGraph.copy() (BVGraph.java:553-578): shares the device data, own stream/workspace."""
        h = C.c_void_p()
        _check(lib().bvg_copy(self._h, C.byref(h)), "copy")
        g = BVGraph(h); g._basename = self._basename
        if self.node_base():
            g.set_node_base(self.node_base())                               # (a flyweight answers in the same id space)
        return g

    def set_node_base(self, base):
        _check(lib().bvg_set_node_base(self._h, base), "set_node_base")
        self._node_base = int(base)

    def node_base(self):
        return getattr(self, "_node_base", 0)

    def set_tuning(self, block_bits=0, force_wide=False, force_slow=False, stream=False, grab_threshold=0, legacy=False, no_index=False):
        # (no_index: False / True, or 2 = "marks only": the index this handle builds keeps the validation marks and entries for lists of >= 4 096 residuals only)
        """stream=True selects the experimental streaming data-flow kernel (experimental/bvg_stream.hip) as tier 0; no_index=True makes
        this handle scan without the residual skip index (neither built nor read)."""
        t = Tuning(block_bits, int(force_wide), int(force_slow), (2 if stream else (1 if legacy else 0)) | (int(grab_threshold) << 8), int(no_index))
        _check(lib().bvg_set_tuning(self._h, C.byref(t)), "set_tuning")

    def offsets(self):
        out = np.empty(self.num_nodes() + 1, dtype=np.uint64)
        _check(lib().bvg_get_offsets(self._h, out.ctypes.data), "get_offsets")
        return out

    def outdegrees(self, frm=0, to=None):
        to = self.num_nodes() if to is None else to
        out = np.empty(max(to - frm, 0), dtype=np.int32)
        _check(lib().bvg_outdegrees(self._h, frm, to, out.ctypes.data if len(out) else None) if to > frm else (0 if 0 <= frm <= self.num_nodes() else _abi.E_ARG), "outdegrees")
        return out

    def outdegree(self, x):
        if x < 0 or x >= self.num_nodes():
            raise IllegalArgumentException(_abi.E_ARG, "outdegree(%d)" % x)               # BVG:823
        return int(self.outdegrees(x, x + 1)[0])

    def decode_range(self, frm, to):
        """(outdeg int32[to-frm], succ int64[sum]) of nodes [frm,to) — bit-exact with nodeIterator(frm)."""
        if frm < 0 or to > self.num_nodes() or frm > to:
            raise IllegalArgumentException(_abi.E_ARG, "decode_range(%d,%d)" % (frm, to))
        cnt = to - frm
        deg = np.empty(max(cnt, 1), dtype=np.int32)
        need = C.c_uint64(0)
        cap = max(1024, 16 * cnt)
        while True:
            succ = np.empty(cap, dtype=np.int64)
            st = lib().bvg_decode_range(self._h, frm, to, deg.ctypes.data, succ.ctypes.data, cap, C.byref(need))
            if st == _abi.E_CAPACITY:
                cap = int(need.value)
                continue
            _check(st, "decode_range(%d,%d)" % (frm, to))
            return deg[:cnt], succ[:need.value]

    def decode_range32(self, frm, to):
        """decode_range with the successors as uint32 (bvg_decode_range32: graphs whose ids fit 32 bits; half the bytes over PCIe)."""
        if frm < 0 or to > self.num_nodes() or frm > to:
            raise IllegalArgumentException(_abi.E_ARG, "decode_range32(%d,%d)" % (frm, to))
        cnt = to - frm
        deg = np.empty(max(cnt, 1), dtype=np.int32)
        need = C.c_uint64(0)
        cap = max(1024, 16 * cnt)
        while True:
            succ = np.empty(cap, dtype=np.uint32)
            st = lib().bvg_decode_range32(self._h, frm, to, deg.ctypes.data, succ.ctypes.data, cap, C.byref(need))
            if st == _abi.E_CAPACITY:
                cap = int(need.value)
                continue
            _check(st, "decode_range32(%d,%d)" % (frm, to))
            return deg[:cnt], succ[:need.value]

    def successors_batch(self, nodes):
        """successors(x) for a frontier: returns (outdeg int32[len(nodes)], succ int64[sum]) in request order."""
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        deg = np.empty(max(len(nodes), 1), dtype=np.int32)
        need = C.c_uint64(0)
        cap = max(1024, 16 * len(nodes))
        while True:
            succ = np.empty(cap, dtype=np.int64)
            st = lib().bvg_successors_batch(self._h, nodes.ctypes.data if len(nodes) else None, len(nodes), deg.ctypes.data, succ.ctypes.data, cap, C.byref(need))
            if st == _abi.E_CAPACITY:
                cap = int(need.value)
                continue
            _check(st, "successors_batch")
            return deg[:len(nodes)], succ[:need.value]

    def successor_array(self, x):
        if x < 0 or x >= self.num_nodes():
            raise IllegalArgumentException(_abi.E_ARG, "successors(%d)" % x)              # BVG:863
        return self.decode_range(x, x + 1)[1]

    successorBigArray = successor_array

    def successors(self, x):
        return LazyLongIterator(self.successor_array(x))

    def node_iterator(self, frm=0):
        return NodeIterator(self, frm)

    nodeIterator = node_iterator

    def split_node_iterators(self, how_many):
        """ImmutableGraph.splitNodeIterators (ImmutableGraph.java:405-436): ceil(n/k)-sized ranges."""
        n = self.num_nodes()
        if how_many <= 0:
            raise IllegalArgumentException(_abi.E_ARG, "splitNodeIterators")
        m = -(-n // how_many) if n else 0
        its = []
        for i in range(how_many):
            lo = min(i * m, n)
            hi = min(lo + m, n)
            its.append(NodeIterator(self.copy(), lo, hi) if lo < n else NodeIterator(self, n, n))
        return its

    splitNodeIterators = split_node_iterators

    def split_by_bits(self, k):
        b = np.empty(k + 1, dtype=np.int64)
        _check(lib().bvg_split_by_bits(self._h, k, b.ctypes.data), "split_by_bits")
        return b

    def split_by_arcs(self, k):
        """Node-range split points with ~equal arc counts (HyperBall.java:748-768 over the cumulative outdegrees)."""
        b = np.empty(k + 1, dtype=np.int64)
        _check(lib().bvg_split_by_arcs(self._h, k, b.ctypes.data), "split_by_arcs")
        return b

    def shard_bounds(self, k, balance=None):
        """bounds[0..k] of the k-way node-range split (bvg_shard_bounds): BALANCE_NODES is ImmutableGraph.java:415-433."""
        b = np.empty(k + 1, dtype=np.int64)
        _check(lib().bvg_shard_bounds(self._h, k, BALANCE_ARCS if balance is None else balance, b.ctypes.data), "shard_bounds")
        return b

    def scan_shard(self, k, r, balance=None):
        """The scan of shard r of k (bvg_scan_shard): dict of bvg_scan_result plus 'from' / 'to'."""
        res = ScanResult(); lo = C.c_int64(); hi = C.c_int64()
        _check(lib().bvg_scan_shard(self._h, k, r, BALANCE_ARCS if balance is None else balance, C.byref(res), C.byref(lo), C.byref(hi)), "scan_shard(%d,%d)" % (k, r))
        d = res.as_dict(); d["from"] = lo.value; d["to"] = hi.value
        return d

    def transpose(self):
        """The transpose in CSR form (toffsets uint64[n+1], tsucc int64[arcs]): the decode + sort of Transform.transposeOffline
        (Transform.java:1058-1160) done on the device; sources of every node's incoming arcs in increasing order."""
        n = self.num_nodes()
        toff = np.empty(n + 1, dtype=np.uint64)
        need = C.c_uint64(0)
        st = lib().bvg_transpose(self._h, toff.ctypes.data, None, 0, C.byref(need))
        if st not in (0, _abi.E_CAPACITY):
            _check(st, "transpose")
        tsucc = np.empty(max(int(need.value), 1), dtype=np.int64)
        _check(lib().bvg_transpose(self._h, toff.ctypes.data, tsucc.ctypes.data, len(tsucc), C.byref(need)), "transpose")
        return toff, tsucc[:int(need.value)]

    def symmetrize(self):
        """The symmetrised graph in CSR form (soffsets uint64[n+1], ssucc int64): Transform.symmetrizeOffline
        (Transform.java:546-575) = union of the graph and its transpose, computed on the device."""
        n = self.num_nodes()
        soff = np.empty(n + 1, dtype=np.uint64)
        need = C.c_uint64(0)
        ssucc = np.empty(max(2 * max(int(self._params.arcs), 0), 1), dtype=np.int64)       # 2 x arcs always suffices
        st = lib().bvg_symmetrize(self._h, soff.ctypes.data, ssucc.ctypes.data, len(ssucc), C.byref(need))
        if st == _abi.E_CAPACITY:                                                          # numArcs unknown in the properties
            ssucc = np.empty(int(need.value), dtype=np.int64)
            st = lib().bvg_symmetrize(self._h, soff.ctypes.data, ssucc.ctypes.data, len(ssucc), C.byref(need))
        _check(st, "symmetrize")
        return soff, ssucc[:int(need.value)]

    def build_index(self, frm=0, to=None):
        """Builds the residual skip index (and validates the blocks) of nodes [frm, to) now (bvg_build_index) instead of inside
        the first scan; returns (entries, bytes) of the graph's index afterwards."""
        to = self.num_nodes() if to is None else to
        e = C.c_uint64(); b = C.c_uint64()
        _check(lib().bvg_build_index(self._h, frm, to, C.byref(e), C.byref(b)), "build_index(%d,%d)" % (frm, to))
        return int(e.value), int(b.value)

    def save_index(self, path=None):
        """Writes the device index (block plan + residual skip index) to `path` (default: basename.bvgidx, which load() picks up)."""
        path = path or (self._basename + ".bvgidx")
        _check(lib().bvg_save_index(self._h, os.fsencode(path)), "save_index(%s)" % path)
        return path

    def load_index(self, path):
        """Loads an index written by save_index(); IOException if the file does not belong to this graph."""
        _check(lib().bvg_load_index(self._h, os.fsencode(path)), "load_index(%s)" % path)

    def scan(self, frm=0, to=None):
        """Full sequential successor scan consumed on chip (SpeedTest.java:127-141): dict of bvg_scan_result."""
        to = self.num_nodes() if to is None else to
        r = ScanResult()
        _check(lib().bvg_scan(self._h, frm, to, C.byref(r)), "scan(%d,%d)" % (frm, to))
        return r.as_dict()


BALANCE_NODES, BALANCE_BITS, BALANCE_ARCS = 0, 1, 2


def mosaic(graphs, cycles):
    """Synthetic workload helper (bvg_mosaic): the cycle of the given base graphs, back to back, repeated `cycles` times."""
    k = len(graphs)
    hs = (C.c_void_p * k)(*[g._h for g in graphs])
    h = C.c_void_p()
    _check(lib().bvg_mosaic(hs, k, cycles, C.byref(h)), "mosaic")
    return BVGraph(h)


def scan_multi(graphs, balance=BALANCE_ARCS):
    """bvg_scan_multi: graphs[i] scans shard i of len(graphs) on its own device, all at once; returns (total, [per shard])."""
    k = len(graphs)
    hs = (C.c_void_p * k)(*[g._h for g in graphs])
    tot = ScanResult(); per = (ScanResult * k)()
    _check(lib().bvg_scan_multi(hs, k, balance, C.byref(tot), per), "scan_multi")
    return tot.as_dict(), [p.as_dict() for p in per]


LABEL_GAMMA_INT, LABEL_FIXED_INT, LABEL_FIXED_INT_LIST, LABEL_FIXED_LONG_LIST = 1, 2, 3, 4


def parse_label_spec(spec):
    """Label.toSpec() text (e.g. "...labelling.FixedWidthIntLabel(FOO,10)") -> (kind, width)."""
    k, w = C.c_int(), C.c_int()
    _check(lib().bvg_labels_parse_spec(spec.encode() if isinstance(spec, str) else spec, C.byref(k), C.byref(w)), "labelspec %r" % (spec,))
    return k.value, w.value


class LabelledArcIterator(LazyLongIterator):
    """ArcLabelledNodeIterator.LabelledArcIterator (labelling/ArcLabelledNodeIterator.java): successors with label()."""

    def __init__(self, succ, labels):
        LazyLongIterator.__init__(self, succ)
        self._l = labels

    def label(self):
        """The label of the arc returned by the last next_long() (BitStreamArcLabelledImmutableGraph.java:250-252)."""
        if self._i == 0:
            raise IllegalStateException(_abi.E_STATE, "label() before nextLong()")
        return int(self._l[self._i - 1])


class BitStreamArcLabelledImmutableGraph:
    """labelling/BitStreamArcLabelledImmutableGraph.java: an underlying BVGraph plus one int label per arc, both decoded on the
    device (GammaCodedIntLabel / FixedWidthIntLabel)."""

    def __init__(self, graph, handle, keep=()):
        self.g = graph
        self._h = handle
        self._keep = keep

    @classmethod
    def load(cls, basename, device=0):
        """load(basename) (:378-484): basename.properties names the underlying graph and the label class."""
        buf = C.create_string_buffer(4096)
        _check(lib().bvg_labels_read_properties(os.fsencode(basename), None, None, buf, len(buf)), "load(%s)" % basename)
        under = os.fsdecode(buf.value)
        g = BVGraph.load(under, device)
        h = C.c_void_p()
        buf = C.create_string_buffer(4096)
        _check(lib().bvg_labels_open(os.fsencode(basename), g.num_nodes(), device, C.byref(h), buf, len(buf)), "load(%s)" % basename)
        return cls(g, h)

    @classmethod
    def from_memory(cls, graph, kind, width, stream, label_offsets, device=0):
        st = np.frombuffer(bytes(stream), dtype=np.uint8) if not isinstance(stream, np.ndarray) else np.ascontiguousarray(stream, dtype=np.uint8)
        lo = np.ascontiguousarray(label_offsets, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().bvg_labels_open_mem(kind, width, graph.num_nodes(), st.ctypes.data if len(st) else None, len(st), lo.ctypes.data, device, C.byref(h)), "labels_open_mem")
        return cls(graph, h)

    def close(self):
        if getattr(self, "_h", None):
            lib().bvg_labels_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_nodes(self):
        return self.g.num_nodes()

    def num_arcs(self):
        return self.g.num_arcs()

    def outdegree(self, x):
        return self.g.outdegree(x)

    def decode_range(self, frm, to):
        """(outdeg, successors, labels) of nodes [frm,to): one batch of the labelled node iterator (:565-582)."""
        deg, succ = self.g.decode_range(frm, to)
        lab = np.empty(max(len(succ), 1), dtype=np.int32)
        n = C.c_uint64()
        d32 = np.ascontiguousarray(deg, dtype=np.int32)
        _check(lib().bvg_labels_decode_range(self._h, frm, to, d32.ctypes.data if len(d32) else None, lab.ctypes.data, len(succ), C.byref(n)), "labels(%d,%d)" % (frm, to))
        return deg, succ, lab[:len(succ)]

    def decode_range_lists(self, frm, to):
        """List labels (FixedWidthIntListLabel: int32 values; FixedWidthLongListLabel: int64): (outdeg, successors, list_off[arcs+1], values)."""
        deg, succ = self.g.decode_range(frm, to)
        d32 = np.ascontiguousarray(deg, dtype=np.int32)
        loff = np.zeros(len(succ) + 1, dtype=np.uint64)
        n = C.c_uint64()
        kind = C.c_int(); width = C.c_int(); nn = C.c_int64(); sb = C.c_uint64()
        _check(lib().bvg_labels_info(self._h, C.byref(kind), C.byref(width), C.byref(nn), C.byref(sb)), "labels_info")
        fn, dt = (lib().bvg_labels_decode_range_lists64, np.int64) if kind.value == LABEL_FIXED_LONG_LIST else (lib().bvg_labels_decode_range_lists, np.int32)
        st = fn(self._h, frm, to, d32.ctypes.data if len(d32) else None, loff.ctypes.data, None, 0, C.byref(n))
        if st != _abi.E_CAPACITY:
            _check(st, "label lists(%d,%d)" % (frm, to))
        vals = np.empty(max(n.value, 1), dtype=dt)
        if n.value:
            _check(fn(self._h, frm, to, d32.ctypes.data, loff.ctypes.data, vals.ctypes.data, n.value, C.byref(n)), "label lists(%d,%d)" % (frm, to))
        return deg, succ, loff, vals[:n.value]

    def successors(self, x):
        """successors(x) (:208-229): a LabelledArcIterator."""
        if x < 0 or x >= self.num_nodes():
            raise IllegalArgumentException(_abi.E_ARG, "successors(%d)" % x)
        _, succ, lab = self.decode_range(x, x + 1)
        return LabelledArcIterator(succ, lab)
