// bvg_device.h — device-side bit-stream primitives and instantaneous-code decoders for gfx950.
//
// Bit order and codes are those of dsiutils' InputBitStream as used by BVGraph.java:627-796
// (MSB-first within each byte; unary = zeros then a one; gamma = unary(msb) + msb bits of x+1;
// zeta_k; delta; nibble; Golomb — SURVEY.md Appendix A.2).  Everything is integer/bit work on the
// scalar+vector ALUs: v_ffbh (count leading zeros), shifts, v_perm (byte swap).  No MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bvgraph_hip.h"

namespace bvg {

typedef uint64_t u64 __attribute__((aligned(1)));   // unaligned 8-byte global load (gfx950 handles it in hardware)

// Error bits accumulated on the device (mapped to bvg_status by the host).
enum : unsigned {
    ERR_REF_RANGE = 1u,     // reference > window or before node 0      -> BVG_E_STATE (BVG:701)
    ERR_OVERRUN = 2u,       // record runs past its end / past the file -> BVG_E_EOF
    ERR_MALFORMED = 4u,     // negative extra count, offsets disagree   -> BVG_E_EOF
    ERR_CAPACITY = 8u,      // slow-path pool too small (host retries with a larger pool)
};

struct Codings {
    int outdegree, block, residual, reference, block_count;
    int zeta_k;
};

// A cursor over the .graph bytes in global memory.  `limit_byte` is the last byte index that may be
// the start of an 8-byte load (the buffer is padded); positions beyond read as the padding zeros.
struct BitCursor {
    const uint8_t* base;
    uint64_t pos;          // absolute bit position
    uint64_t limit_byte;
    // Optional LDS window over the stream: win[i & win_mask] = big-endian dword i of the staged bytes,
    // dword 0 starting at absolute bit win_bit0; bits [win_lo, win_hi) (relative to win_bit0) are valid.
    // A linear window has win_mask = ~0u, win_lo = 0; a ring keeps the last (win_mask+1) dwords.
    // Positions outside the window fall back to global memory.
    const uint32_t* win = nullptr;
    uint64_t win_bit0 = 0;
    uint32_t win_hi = 0;
    uint32_t win_lo = 0;
    uint32_t win_mask = 0xFFFFFFFFu;

    __device__ __forceinline__ uint64_t peek_global() const {
        uint64_t byte = pos >> 3;
        byte = byte < limit_byte ? byte : limit_byte;
        const uint8_t* p = base + byte;
        uint64_t hi = __builtin_bswap64(*reinterpret_cast<const u64*>(p));
        unsigned sh = (unsigned)pos & 7u;
        // 57..64 valid bits; the 9th byte completes the window
        uint64_t w = hi << sh;
        if (sh) w |= (uint64_t)p[8] >> (8u - sh);
        return w;
    }
    __device__ __forceinline__ uint64_t peek() const {
        const uint64_t rel = pos - win_bit0;
        if (rel >= (uint64_t)win_lo && rel + 96 <= (uint64_t)win_hi) {   // (wraps to huge when pos < win_bit0)
            const uint32_t wi = (uint32_t)rel >> 5, sh = (uint32_t)rel & 31u;
            const uint32_t a = win[wi & win_mask], b = win[(wi + 1) & win_mask], c = win[(wi + 2) & win_mask];
            const uint64_t ab = ((uint64_t)a << 32) | b;
            return sh ? (ab << sh) | (uint64_t)(c >> (32u - sh)) : ab;
        }
        return peek_global();
    }
    __device__ __forceinline__ void skip(unsigned n) { pos += n; }

    // n <= 64 bits, MSB first.
    __device__ __forceinline__ uint64_t read_bits(unsigned n) {
        if (n == 0) return 0;
        uint64_t w = peek();
        pos += n;
        return w >> (64u - n);
    }
    // Unary: zeros before the first one.  Long runs (>= 64 zeros) loop; bounded by the padding (zeros
    // forever would spin), so the caller's record-end check plus `guard` stops runaway streams.
    __device__ __forceinline__ uint64_t read_unary(uint64_t guard_pos) {
        uint64_t z = 0;
        for (;;) {
            uint64_t w = peek();
            if (w) { unsigned lz = (unsigned)__builtin_clzll(w); pos += lz + 1; return z + lz; }
            pos += 64; z += 64;
            if (pos > guard_pos) return z;   // corrupt: let the caller flag the overrun
        }
    }
    __device__ __forceinline__ uint64_t read_gamma(uint64_t guard_pos) {
        uint64_t w = peek();
        unsigned lz = w ? (unsigned)__builtin_clzll(w) : 64u;
        if (lz < 32) {                      // whole code inside the window: 2*lz+1 <= 63 bits
            unsigned len = 2 * lz + 1;
            pos += len;
            return (w >> (64u - len)) - 1;
        }
        uint64_t msb = read_unary(guard_pos);
        if (msb > 63) { pos = guard_pos + 1; return 0; }
        return ((1ull << msb) | read_bits((unsigned)msb)) - 1;
    }
    __device__ __forceinline__ uint64_t read_delta(uint64_t guard_pos) {
        uint64_t msb = read_gamma(guard_pos);
        if (msb > 63) { pos = guard_pos + 1; return 0; }
        return ((1ull << msb) | read_bits((unsigned)msb)) - 1;
    }
    __device__ __forceinline__ uint64_t read_zeta(unsigned k, uint64_t guard_pos) {
        uint64_t w = peek();
        unsigned h = w ? (unsigned)__builtin_clzll(w) : 64u;
        unsigned nb = h * k + k - 1;                    // payload bits before the optional extra bit
        if (h + 1 + nb + 1 <= 64) {
            uint64_t t = nb ? ((w << (h + 1)) >> (64u - nb)) : 0;
            uint64_t left = 1ull << (h * k);
            if (t < left) { pos += h + 1 + nb; return t + left - 1; }
            uint64_t bit = (w >> (64u - (h + 1 + nb + 1))) & 1u;
            pos += h + 1 + nb + 1;
            return (t << 1) + bit - 1;
        }
        uint64_t hh = read_unary(guard_pos);
        if (hh * k + k - 1 > 63) { pos = guard_pos + 1; return 0; }
        uint64_t left = 1ull << (hh * k);
        uint64_t m = read_bits((unsigned)(hh * k + k - 1));
        if (m < left) return m + left - 1;
        return (m << 1) + read_bits(1) - 1;
    }
    __device__ __forceinline__ uint64_t read_nibble(uint64_t guard_pos) {
        uint64_t x = 0, stop;
        do { x <<= 3; uint64_t g = read_bits(4); stop = g >> 3; x |= g & 7; } while (!stop && pos <= guard_pos);
        return x;
    }
    __device__ __forceinline__ uint64_t read_golomb(uint64_t m, uint64_t guard_pos) {
        if (m == 0) return 0;
        uint64_t q = read_unary(guard_pos);
        if (m == 1) return q;
        unsigned l = 63u - (unsigned)__builtin_clzll(m);
        uint64_t thr = (1ull << (l + 1)) - m;
        uint64_t x = read_bits(l);
        if (x >= thr) x = ((x << 1) + read_bits(1)) - thr;
        return q * m + x;
    }
    // Field dispatch: `coding` is uniform across the grid (a property of the file), so these
    // branches never diverge.
    __device__ __forceinline__ uint64_t read_coded(int coding, unsigned k, uint64_t guard_pos) {
        switch (coding) {
            case BVG_GAMMA: return read_gamma(guard_pos);
            case BVG_ZETA: return read_zeta(k, guard_pos);
            case BVG_UNARY: return read_unary(guard_pos);
            case BVG_DELTA: return read_delta(guard_pos);
            case BVG_NIBBLE: return read_nibble(guard_pos);
            case BVG_GOLOMB: return read_golomb(k, guard_pos);
        }
        return 0;
    }
};

__device__ __forceinline__ int64_t nat2int(uint64_t u) { return (u & 1) ? -(int64_t)((u + 1) >> 1) : (int64_t)(u >> 1); }

// ---- checksum (definition in include/bvgraph_hip.h) ----
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// one arc: k1 * y + k0 (mod 2^64) -- linear in y under a per-node odd key, so a kernel pays ONE v_mad_u64_u32 per arc (acc = y * k1 + acc) and
// adds the d * k0 of a node once; a single wrong, missing or surplus successor always changes the sum
// the per-node key of the scan checksum (include/bvgraph_hip.h): a multiply / xor-shift hash of the 64-bit node id -- ten 32-bit vector
// instructions, so a kernel recomputes k1 wherever a task needs it instead of carrying it around (splitmix64, rounds 1-4, is ~25)
__host__ __device__ __forceinline__ void node_key(uint64_t x, uint32_t& k0, uint32_t& k1) {
    uint32_t h = (uint32_t)x * 0x9E3779B1u + (uint32_t)(x >> 32) * 0x85EBCA77u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    k1 = h | 1u;
    k0 = h * 0x297A2D39u; k0 ^= k0 >> 15;
}
__host__ __device__ __forceinline__ uint64_t mix_keyed(uint32_t k0, uint32_t k1, uint64_t y) { return (uint64_t)k1 * y + (uint64_t)k0; }
// what a node of outdegree d adds besides k1 * (its successors relative to `nbase`): d * (k1 * nbase + k0)
__host__ __device__ __forceinline__ uint64_t mix_node_const(uint32_t k0, uint32_t k1, uint64_t nbase, uint64_t d) { return d * ((uint64_t)k1 * nbase + (uint64_t)k0); }

// ---- wave64 helpers ----
__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
// Cross-lane data movement with DPP (no LDS traffic): lanes without a source, or masked off, read 0.
template <int CTRL, int ROW_MASK, int BANK_MASK> __device__ __forceinline__ uint32_t dpp0(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
constexpr int kRowShr = 0x110, kRowBcast15 = 0x142, kRowBcast31 = 0x143;
// value of lane `idx` (wave-uniform idx) broadcast through an SGPR
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t idx) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)idx));
}
__device__ __forceinline__ uint64_t lane_get64(uint64_t v, uint32_t idx) {
    return ((uint64_t)lane_get((uint32_t)(v >> 32), idx) << 32) | lane_get((uint32_t)v, idx);
}
// inclusive prefix sum over the 64 lanes of a wavefront: 4 rows of 16 lanes scanned with row_shr, then
// stitched with the two row broadcasts
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    uint32_t v = x + dpp0<kRowShr + 1, 0xf, 0xf>(x) + dpp0<kRowShr + 2, 0xf, 0xf>(x) + dpp0<kRowShr + 3, 0xf, 0xf>(x);
    v += dpp0<kRowShr + 4, 0xf, 0xe>(v);
    v += dpp0<kRowShr + 8, 0xf, 0xc>(v);
    v += dpp0<kRowBcast15, 0xa, 0xf>(v);
    v += dpp0<kRowBcast31, 0xc, 0xf>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) { return lane_get(wave_incl_scan(v), 63); }
__device__ __forceinline__ uint32_t wave_max32(uint32_t x) {
    uint32_t v = max(max(x, dpp0<kRowShr + 1, 0xf, 0xf>(x)), max(dpp0<kRowShr + 2, 0xf, 0xf>(x), dpp0<kRowShr + 3, 0xf, 0xf>(x)));
    v = max(v, dpp0<kRowShr + 4, 0xf, 0xe>(v));
    v = max(v, dpp0<kRowShr + 8, 0xf, 0xc>(v));
    v = max(v, dpp0<kRowBcast15, 0xa, 0xf>(v));
    v = max(v, dpp0<kRowBcast31, 0xc, 0xf>(v));
    return lane_get(v, 63);
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_or32(uint32_t x) {
    uint32_t v = x | dpp0<kRowShr + 1, 0xf, 0xf>(x) | dpp0<kRowShr + 2, 0xf, 0xf>(x) | dpp0<kRowShr + 3, 0xf, 0xf>(x);
    v |= dpp0<kRowShr + 4, 0xf, 0xe>(v);
    v |= dpp0<kRowShr + 8, 0xf, 0xc>(v);
    v |= dpp0<kRowBcast15, 0xa, 0xf>(v);
    v |= dpp0<kRowBcast31, 0xc, 0xf>(v);
    return lane_get(v, 63);
}

}  // namespace bvg
