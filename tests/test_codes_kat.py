"""Known-answer vectors for the instantaneous codes BVGraph can be stored with (BVGraph.java:627-796 reads every field through dsiutils'
InputBitStream, which is not in the reference tree): the bit strings below are copied from the PUBLISHED tables of the codes, not
produced by any writer of this repository, so they pin the oracle's readers (CPU) and the HIP decoders (GPU, through the C ABI) on
something outside the three restatements that otherwise only agree with each other.

  * unary as dsiutils writes it: x zeros, then a one;
  * Elias gamma and delta of x + 1 (P. Elias, "Universal codeword sets and representations of the integers", 1975, Table I/II:
    1 -> 1, 2 -> 010, 3 -> 011, 4 -> 00100 ...; delta: 1 -> 1, 2 -> 0100, 3 -> 0101, 4 -> 01100, 8 -> 00100000 ...);
  * zeta_k of x + 1 (P. Boldi, S. Vigna, "Codes for the World Wide Web", Internet Mathematics 2(4), 2005, Table 1: the codes of
    1..8 for k = 1..4; zeta_1 is gamma);
  * Golomb with modulus b (S. Golomb, "Run-length encodings", 1966): quotient in unary (dsiutils' unary: zeros, then a one), remainder
    in minimal binary -- b = 3: 0 -> 1 0, 1 -> 1 10, 2 -> 1 11, 3 -> 01 0 ...; b = 4 (Rice): 0 -> 1 00, 5 -> 01 01;
  * nibble coding as dsiutils' OutputBitStream.writeNibble writes it (restated from its source: 3-bit groups, most significant
    first, each preceded by a bit that is 1 on the LAST group; 0 is 1000) -- no published table exists for it: weaker than the others.
"""
import ctypes as C

import numpy as np
import pytest

# value -> code, most significant bit first
UNARY = {0: "1", 1: "01", 2: "001", 3: "0001", 7: "00000001"}
GAMMA = {0: "1", 1: "010", 2: "011", 3: "00100", 4: "00101", 5: "00110", 6: "00111", 7: "0001000", 8: "0001001", 15: "000010000", 16: "000010001"}
DELTA = {0: "1", 1: "0100", 2: "0101", 3: "01100", 4: "01101", 5: "01110", 6: "01111", 7: "00100000", 8: "00100001", 15: "001010000", 16: "001010001"}
ZETA = {
    1: {0: "1", 1: "010", 2: "011", 3: "00100", 4: "00101", 5: "00110", 6: "00111", 7: "0001000"},
    2: {0: "10", 1: "110", 2: "111", 3: "01000", 4: "01001", 5: "01010", 6: "01011", 7: "011000"},
    3: {0: "100", 1: "1010", 2: "1011", 3: "1100", 4: "1101", 5: "1110", 6: "1111", 7: "0100000"},
    4: {0: "1000", 1: "10010", 2: "10011", 3: "10100", 4: "10101", 5: "10110", 6: "10111", 7: "11000"},
}
GOLOMB = {3: {0: "10", 1: "110", 2: "111", 3: "010", 4: "0110", 5: "0111", 6: "0010", 7: "00110"},
          4: {0: "100", 1: "101", 2: "110", 3: "111", 4: "0100", 5: "0101", 9: "00101"}}
NIBBLE = {0: "1000", 1: "1001", 5: "1101", 7: "1111", 8: "00011000", 63: "01111111", 64: "000100001000"}


def _bytes(bits):
    bits = bits + "0" * (-len(bits) % 8)
    return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))


def _tables():
    yield "unary", 0, UNARY
    yield "gamma", 0, GAMMA
    yield "delta", 0, DELTA
    for k, t in ZETA.items():
        yield "zeta", k, t
    for b, t in GOLOMB.items():
        yield "golomb", b, t
    yield "nibble", 0, NIBBLE


@pytest.mark.parametrize("name,k,table", list(_tables()), ids=lambda v: str(v) if not isinstance(v, dict) else "")
def test_oracle_reads_the_published_codes(oracle, name, k, table):
    """Every code alone, and all of them back to back in one stream (the reader must stop exactly at each boundary)."""
    L = oracle.lib()
    vals = sorted(table)
    stream = "".join(table[v] for v in vals)
    for data, want in [(_bytes(table[v]), [v]) for v in vals] + [(_bytes(stream), vals)]:
        buf = np.frombuffer(data + b"\0" * 16, dtype=np.uint8)
        b = oracle.Bits()
        L.bvgo_bits_init(C.byref(b), buf.ctypes.data, len(data), 0)
        pos = 0
        for v in want:
            got = {"unary": lambda: L.bvgo_read_unary(C.byref(b)), "gamma": lambda: L.bvgo_read_gamma(C.byref(b)), "delta": lambda: L.bvgo_read_delta(C.byref(b)),
                   "zeta": lambda: L.bvgo_read_zeta(C.byref(b), k), "golomb": lambda: L.bvgo_read_golomb(C.byref(b), k), "nibble": lambda: L.bvgo_read_nibble(C.byref(b))}[name]()
            assert got == v, (name, k, v, got)
            pos += len(table[v])
            assert b.pos == pos, (name, k, v)


def _record(values, res_table, d_code):
    """One reference-free record (window 0, no intervals) whose residual codes are the table entries of `values`, in order:
    outdegree, then the first residual as nat2int(v0) (BVG:914) and the others as gaps - 1 (BVG:921).  Returns (bits, successors)."""
    succ = []
    for i, v in enumerate(values):
        if i == 0:
            succ.append(v // 2 if v % 2 == 0 else -(v + 1) // 2)                # Fast.nat2int; node 0: only non-negative first successors are legal
        else:
            succ.append(succ[-1] + v + 1)
    return d_code[len(values)] + "".join(res_table[v] for v in values), succ


CODING = {"gamma": 2, "delta": 1, "zeta": 6, "golomb": 3, "nibble": 7}       # CompressionFlags constants (BVGraph.java:236-283)


@pytest.mark.gpu
@pytest.mark.parametrize("name,k,table", [t for t in _tables() if t[0] != "unary"], ids=lambda v: str(v) if not isinstance(v, dict) else "")
@pytest.mark.parametrize("slow", [False, True])
def test_hip_decoders_read_the_published_codes(W, name, k, table, slow):
    """The same bit strings as the residuals of a hand-assembled one-node graph, decoded through the C ABI: by the LDS decoders of the row
    kernels and by the generic BitCursor reader (force_slow).  The outdegree is gamma coded from the table above."""
    vals = [v for v in sorted(table) if v % 2 == 0][:1] + [v for v in sorted(table)]     # an even value first: a non-negative first successor
    vals = vals[:len(vals) if len(vals) in GAMMA else max(x for x in GAMMA if x <= len(vals))]
    bits, succ = _record(vals, table, GAMMA)
    data = _bytes(bits)
    p = W.default_params(window_size=0, max_ref_count=0, min_interval_length=0, residual_coding=CODING[name], zeta_k=k if k else 3).clone(nodes=max(succ) + 1, arcs=len(succ))
    n = p.nodes
    # node 0 holds the record; the other nodes are empty (outdegree 0 = gamma "1": one bit each)
    full = bits + "1" * (n - 1)
    offs = np.array([0, len(bits)] + [len(bits) + i for i in range(1, n)], dtype=np.uint64)
    g = W.BVGraph.from_memory(p, np.frombuffer(_bytes(full), dtype=np.uint8), offs)
    if slow:
        g.set_tuning(force_slow=True)
    deg, got = g.decode_range(0, n)
    assert deg[0] == len(succ) and deg[1:].sum() == 0
    assert got.tolist() == succ, (name, k)
    g.close()
    assert len(data) > 0
