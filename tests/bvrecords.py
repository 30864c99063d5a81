"""Test helper: hand-assembled BVGraph records (a raw bit writer) and a literal pure-Python restatement of the
reference's lazy iterators, so streams that the encoder never writes can be fed to the oracle and to the HIP path.

The iterators follow /root/reference/src/it/unimi/dsi/big/webgraph line by line (small cases only):
  MaskedLongIterator.java:67-128, MergedLongIterator.java:54-92, LongIntervalSequenceIterator.java:57-95,
  BVGraph.java:995-1097 (successors), :902-935 (ResidualLongIterator).
The writer produces the default codings only (gamma outdegree / block count / blocks / intervals, unary reference,
zeta_k residuals: BVGraph.java:527-542; code definitions SURVEY Appendix A.2).
"""
import numpy as np


class PyBits:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)

    def unary(self, x):
        self.bits += [0] * x + [1]

    def gamma(self, x):
        v = x + 1; b = v.bit_length() - 1
        self.unary(b); self.put(v & ((1 << b) - 1), b)

    def zeta(self, x, k):
        v = x + 1; h = (v.bit_length() - 1) // k; left = 1 << (h * k)
        self.unary(h)
        if v - left < left:
            self.put(v - left, h * k + k - 1)
        else:
            self.put(v, h * k + k)

    def __len__(self):
        return len(self.bits)

    def tobytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


def int2nat(v):
    return 2 * v if v >= 0 else -2 * v - 1


class Record:
    """One node's record, field by field.  Nothing is checked: that is the point."""

    def __init__(self, d, ref=0, blocks=(), intervals=(), residuals=()):
        self.d, self.ref, self.blocks, self.intervals, self.residuals = d, ref, list(blocks), list(intervals), list(residuals)

    def write(self, w, x, window, min_interval, zeta_k, extra_count):
        """extra_count: what the decoder will compute (d - copied); decides which optional sections exist."""
        w.gamma(self.d)
        if self.d == 0:
            return
        if window > 0:
            w.unary(self.ref)
        if self.ref > 0:
            w.gamma(len(self.blocks))
            for i, b in enumerate(self.blocks):
                w.gamma(b if i == 0 else b - 1)
        if extra_count > 0 and min_interval != 0:
            w.gamma(len(self.intervals))
            prev = 0
            for i, (left, ln) in enumerate(self.intervals):
                w.gamma(int2nat(left - x) if i == 0 else left - prev - 1)
                w.gamma(ln - min_interval)
                prev = left + ln
        prev = None
        for i, r in enumerate(self.residuals):
            w.zeta(int2nat(r - x) if i == 0 else r - prev - 1, zeta_k)
            prev = r


# ---- literal iterators ----------------------------------------------------------------------------------------------
class ArrayIt:                                           # LazyLongIterators.wrap(array, n)
    def __init__(self, a):
        self.a, self.i = list(a), 0

    def next(self):
        if self.i >= len(self.a):
            return -1
        self.i += 1
        return self.a[self.i - 1]

    def skip(self, n):
        k = min(n, len(self.a) - self.i); self.i += k
        return k


class IntervalIt:                                        # LongIntervalSequenceIterator.java:57-78
    def __init__(self, left, ln):
        self.left, self.len, self.rem, self.ci, self.idx = list(left), list(ln), len(left), 0, 0
        self.cl = left[0] if left else 0

    def next(self):
        if self.rem == 0:
            return -1
        v = self.cl + self.idx; self.idx += 1
        if self.idx == self.len[self.ci]:
            self.rem -= 1
            if self.rem:
                self.ci += 1; self.cl = self.left[self.ci]
            self.idx = 0
        return v


class MaskedIt:                                          # MaskedLongIterator.java:67-100
    def __init__(self, mask, under):
        self.mask, self.n, self.cur, self.under = list(mask), len(mask), 0, under
        if self.n:
            self.left = self.mask[self.cur]; self.cur += 1
            self._advance()
        else:
            self.left = -1

    def _advance(self):
        if self.left == 0 and self.cur < self.n:
            self.under.skip(self.mask[self.cur]); self.cur += 1
            if self.cur < self.n:
                self.left = self.mask[self.cur]; self.cur += 1
            else:
                self.left = -1

    def next(self):
        if self.left == 0:
            return -1
        v = self.under.next()
        if self.left == -1 or v == -1:
            return v
        if self.left > 0:
            self.left -= 1
            self._advance()
        return v


class MergedIt:                                          # MergedLongIterator.java:54-92
    def __init__(self, it0, it1, n=(1 << 31) - 1):
        self.it0, self.it1, self.n = it0, it1, n
        self.c0, self.c1 = it0.next(), it1.next()

    def next(self):
        if self.n == 0 or (self.c0 == -1 and self.c1 == -1):
            return -1
        self.n -= 1
        if self.c0 == -1:
            r = self.c1; self.c1 = self.it1.next()
        elif self.c1 == -1:
            r = self.c0; self.c0 = self.it0.next()
        elif self.c0 < self.c1:
            r = self.c0; self.c0 = self.it0.next()
        elif self.c0 > self.c1:
            r = self.c1; self.c1 = self.it1.next()
        else:
            r = self.c0; self.c0 = self.it0.next(); self.c1 = self.it1.next()
        return r


def reference_successors(rec, x, lists, min_interval):
    """BVGraph.java:1012-1090 on an already parsed record; `lists[y]` = the d(y) values the reference holds in its window for
    node y (with the -1 padding a deficient list carries).  Returns d values, -1 after the iterator is exhausted."""
    d = rec.d
    if d == 0:
        return []
    extra = d
    if rec.ref > 0:
        total = sum(rec.blocks); copied = sum(rec.blocks[0::2])
        if len(rec.blocks) % 2 == 0:
            copied += len(lists[x - rec.ref]) - total                                  # :1030
        extra = d - copied
    iv = rec.intervals if (extra > 0 and min_interval != 0) else []
    for _, ln in iv:
        extra -= ln
    assert extra == len(rec.residuals), "test record inconsistent: the decoder would read %d residuals" % extra
    assert extra >= 0
    resid = ArrayIt(rec.residuals) if extra else None
    if iv:
        ivit = IntervalIt([l for l, _ in iv], [n for _, n in iv])
        ext = MergedIt(ivit, resid) if resid is not None else ivit
    else:
        ext = resid
    if rec.ref <= 0:
        it = ext
    else:
        # the window holds plain arrays: LazyLongIterators.wrap(window[refIndex], outd[refIndex]) yields their -1 too
        blk = MaskedIt(rec.blocks, ArrayIt(lists[x - rec.ref]))
        it = blk if ext is None else MergedIt(blk, ext, d)
    return [it.next() for _ in range(d)]


def assemble(records, window=7, min_interval=4, zeta_k=3, max_ref=3):
    """records: list of Record (node i = records[i]).  Returns (graph_bytes, offsets uint64[n+1], expected lists)."""
    w = PyBits()
    offs = [0]
    lists = []
    for x, rec in enumerate(records):
        extra = rec.d
        if rec.d and rec.ref > 0:
            total = sum(rec.blocks); copied = sum(rec.blocks[0::2])
            if len(rec.blocks) % 2 == 0:
                copied += len(lists[x - rec.ref]) - total
            extra = rec.d - copied
        rec.write(w, x, window, min_interval, zeta_k, extra)
        offs.append(len(w))
        lists.append(reference_successors(rec, x, lists, min_interval))
    return w.tobytes(), np.array(offs, dtype=np.uint64), lists
