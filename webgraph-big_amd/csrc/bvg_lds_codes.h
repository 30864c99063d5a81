// bvg_lds_codes.h — lean instantaneous-code decoders over an LDS-resident copy of the .graph stream.
//
// The stream is staged as big-endian dwords (byte-swapped once at staging), so an MSB-first window at
// bit position `rel` (relative to the staged base) is two or three ds_read_b32 plus funnel shifts.
// MASK = 0xFFFFFFFF for a linear window, (words-1) for a ring.  Codes as in SURVEY.md Appendix A.2.
#pragma once
#include "bvg_device.h"

namespace bvg {

__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh) {      // bits [sh, sh+32) of hi:lo, sh in 0..31
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (32u - sh));
}
template <uint32_t MASK> __device__ __forceinline__ uint32_t win32(const uint32_t* sring, uint32_t rel) {
    const uint32_t wi = rel >> 5, sh = rel & 31u;
    return funnel(sring[wi & MASK], sring[(wi + 1) & MASK], sh);
}
template <uint32_t MASK> __device__ __forceinline__ uint64_t win64(const uint32_t* sring, uint32_t rel) {
    const uint32_t wi = rel >> 5, sh = rel & 31u;
    const uint32_t a = sring[wi & MASK], b = sring[(wi + 1) & MASK], c = sring[(wi + 2) & MASK];
    return ((uint64_t)funnel(a, b, sh) << 32) | funnel(b, c, sh);
}
// win32 for rel >= 1 (every code but the first of a window: a residual code is never the first code of its record), as ONE v_alignbit_b32 (round 5).  The 32 bits from
// bit `rel` on END in dword wj = (rel + 31) >> 5, 31 - ((rel + 31) & 31) bits above its low end: {dword wj - 1, dword wj} shifted right by ~(rel + 31) & 31, a shift of
// 0 ... 31 that the instruction takes as it is.  (The plain form {dword wi, dword wi + 1} >> (32 - sh) needs a shift of 32 at sh = 0: the compiler turns it into
// v_pk_mov + v_lshrrev_b64.)
template <uint32_t MASK> __device__ __forceinline__ uint32_t win32p(const uint32_t* sring, uint32_t rel) {
    const uint32_t q = rel + 31u, wj = q >> 5;
    return __builtin_amdgcn_alignbit(sring[(wj - 1u) & MASK], sring[wj & MASK], ~q);
}
// gamma from a 64-bit window: value < 2^31 (length <= 63); returns length, 0 = does not fit
__device__ __forceinline__ uint32_t gamma64(uint64_t w, uint64_t& val) {
    const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
    const uint32_t len = 2 * lz + 1;
    val = lz < 32 ? (w >> (64u - len)) - 1 : 0;
    return lz < 32 ? len : 0u;
}
// zeta_k from a 64-bit window; returns length, 0 = does not fit
__device__ __forceinline__ uint32_t zeta64(uint64_t w, uint32_t k, uint64_t& val) {
    const uint32_t h = w ? (uint32_t)__builtin_clzll(w) : 64u;
    const uint32_t nb = h * k + k - 1, zt = h + 1 + nb;
    if (zt + 1 > 64) { val = 0; return 0; }
    const uint64_t t = nb ? ((w << (h + 1)) >> (64u - nb)) : 0;
    const uint64_t left = 1ull << (h * k);
    if (t < left) { val = t + left - 1; return zt; }
    val = ((t << 1) | ((w >> (63u - zt)) & 1ull)) - 1;
    return zt + 1;
}
__device__ __forceinline__ int64_t nat2int64(uint64_t u) { return (int64_t)(u >> 1) ^ -(int64_t)(u & 1); }

// Generic field decode for non-default codings (GEN) from the LDS ring; returns length, 0 = fail over.
// (takes the 64-bit window, so it is independent of how the stream is staged)
static __device__ __noinline__ uint32_t decode_generic_w(uint64_t w, int coding, uint32_t k, uint64_t* out) {
    uint64_t v = 0; uint32_t len = 0;
    switch (coding) {
        case BVG_GAMMA: len = gamma64(w, v); break;
        case BVG_ZETA: len = zeta64(w, k, v); break;
        case BVG_UNARY: { const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u; v = lz; len = lz < 64 ? lz + 1 : 0; break; }
        case BVG_DELTA: {
            uint64_t msb; const uint32_t l1 = gamma64(w, msb);
            if (l1 && msb < 32 && l1 + msb <= 64) { v = ((1ull << msb) | (msb ? (w << l1) >> (64u - msb) : 0)) - 1; len = l1 + (uint32_t)msb; }
            break;
        }
        case BVG_NIBBLE: {
            uint32_t used = 0; uint64_t x = 0; bool stop = false;
            while (!stop && used + 4 <= 64) { const uint32_t g = (uint32_t)(w >> (60u - used)) & 15u; x = (x << 3) | (g & 7u); stop = g >> 3; used += 4; }
            if (stop) { v = x; len = used; }
            break;
        }
        case BVG_GOLOMB: {
            const uint32_t q = w ? (uint32_t)__builtin_clzll(w) : 64u;
            if (k == 0) { v = 0; len = 0; break; }
            if (q < 40) {
                if (k == 1) { v = q; len = q + 1; break; }
                const uint32_t l = 31u - (uint32_t)__builtin_clz(k); const uint32_t thr = (1u << (l + 1)) - k;
                const uint64_t rest = w << (q + 1);
                uint32_t xr = l ? (uint32_t)(rest >> (64u - l)) : 0; uint32_t used = q + 1 + l;
                if (xr >= thr) { xr = ((xr << 1) | (uint32_t)((rest >> (63u - l)) & 1ull)) - thr; used++; }
                v = (uint64_t)q * k + xr; len = used;
            }
            break;
        }
    }
    *out = v;
    return len;
}

}  // namespace bvg
