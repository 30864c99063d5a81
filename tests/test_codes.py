"""CPU: the instantaneous codes of the oracle against an independent pure-Python writer/reader
 and against the tooling encoder's writers."""
import numpy as np
import pytest


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

    def delta(self, x):
        v = x + 1; b = v.bit_length() - 1
        self.gamma(b); self.put(v & ((1 << b) - 1), b)

    def zeta(self, x, k):
        v = x + 1; h = (v.bit_length() - 1) // k; left = 1 << (h * k)
        self.unary(h)
        if v - left < left:
            self.put(v - left, h * k + k - 1)
        else:
            self.put(v, h * k + k)

    def tobytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


VALUES = [0, 1, 2, 3, 4, 7, 8, 15, 16, 31, 62, 63, 64, 255, 256, 1000, 65535, 65536, 1 << 20, (1 << 31) - 1, 1 << 31, (1 << 40) + 12345, (1 << 62) - 1]


def _reader(O, data):
    import ctypes as C
    buf = np.frombuffer(data + b"\0" * 16, dtype=np.uint8)
    b = O.Bits()
    O.lib().bvgo_bits_init(C.byref(b), buf.ctypes.data, len(data), 0)
    return b, buf


@pytest.mark.parametrize("code", ["unary", "gamma", "delta", "zeta1", "zeta3", "zeta5", "zeta7"])
def test_oracle_reads_python_written_codes(oracle, code):
    import ctypes as C
    vals = [v for v in VALUES if not (code == "unary" and v > 5000) and not (code.startswith("zeta") and v >= 1 << 48)]
    w = PyBits()
    for v in vals:
        {"unary": w.unary, "gamma": w.gamma, "delta": w.delta}.get(code, lambda x: w.zeta(x, int(code[4:])))(v)
    b, keep = _reader(oracle, w.tobytes())
    L = oracle.lib()
    for v in vals:
        got = {"unary": L.bvgo_read_unary, "gamma": L.bvgo_read_gamma, "delta": L.bvgo_read_delta}.get(code, None)
        r = got(C.byref(b)) if got else L.bvgo_read_zeta(C.byref(b), int(code[4:]))
        assert r == v, (code, v, r)
    assert b.err == 0 and b.pos == len(w.bits)


@pytest.mark.parametrize("coding,k", [(1, 0), (2, 0), (5, 0), (6, 1), (6, 3), (6, 4), (7, 0), (3, 1), (3, 5), (3, 8), (3, 13)])
def test_oracle_reads_tooling_written_codes(oracle, tools, coding, k):
    """Encoder writers (tools/bvg_store.cpp) and oracle readers are independent restatements of dsiutils."""
    import ctypes as C
    vals = [v for v in VALUES if not ((coding == 5 or coding == 3) and v > 20000) and not (coding == 6 and v >= 1 << 48)]
    data = tools.encode_values(vals, coding, k).tobytes()
    b, keep = _reader(oracle, data)
    L = oracle.lib()
    for v in vals:
        if coding == 1: r = L.bvgo_read_delta(C.byref(b))
        elif coding == 2: r = L.bvgo_read_gamma(C.byref(b))
        elif coding == 5: r = L.bvgo_read_unary(C.byref(b))
        elif coding == 6: r = L.bvgo_read_zeta(C.byref(b), k)
        elif coding == 7: r = L.bvgo_read_nibble(C.byref(b))
        else: r = L.bvgo_read_golomb(C.byref(b), k)
        assert r == v, (coding, k, v, r)
    assert b.err == 0


def test_zeta3_lengths_match_fixture_histogram_support(tools):
    """zeta_3 code lengths are 3,4,7,8,11,12,...."""
    lens = set()
    for v in [0, 1, 2, 3, 6, 7, 8, 55, 56, 63, 64, 511, 512]:
        n = len(tools.encode_values([v] * 8, 6, 3)) * 8 // 8
        lens.add(n)
    assert lens <= {3, 4, 7, 8, 11, 12, 15, 16}


def test_nat2int(oracle):
    L = oracle.lib()
    assert [L.bvgo_nat2int(u) for u in range(7)] == [0, -1, 1, -2, 2, -3, 3]


def test_eof_is_reported(oracle):
    import ctypes as C
    b, keep = _reader(oracle, b"\x00\x00")
    oracle.lib().bvgo_read_gamma(C.byref(b))
    assert b.err != 0
