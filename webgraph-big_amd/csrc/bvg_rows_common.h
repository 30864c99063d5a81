// bvg_rows_common.h — helpers shared by the row kernels (bvg_rows.hip: one wavefront per block; bvg_rows_wg.hip: one
// workgroup of several wavefronts per block).
#pragma once
#include "bvg_kernels.h"
#include "bvg_lds_codes.h"

namespace bvg {
namespace rows {

constexpr uint32_t kInf = 0xFFFFFFFFu;
constexpr uint32_t LIN = 0xFFFFFFFFu;          // linear window: no index mask
constexpr uint32_t RM = kRing - 1;
constexpr uint32_t kMinTask = 4;               // shortest task (outputs) worth a seek

template <typename T> __device__ __forceinline__ T sentinel() { return (T)~(T)0; }

// XCD-aware work order.  The hardware deals workgroups to the 8 XCDs of an MI355X round robin (workgroup i runs on XCD i mod 8), and
// every XCD has an L2 of its own.  Consecutive node blocks share what lies at their seam -- the 128-byte line of the stream the boundary
// falls into, the halo records the later block decodes again, the lines of the offsets and of the skip entries -- so they should meet in
// ONE L2: workgroup i takes position (i mod 8) * ceil-share + i / 8 of the work list, i.e. each XCD walks its own contiguous eighth of the
// list in order (a bijection of [0, G) for every G).  `xcds` = 1 restores the plain order (experiments).
__device__ __forceinline__ uint32_t xcd_order(uint32_t i, uint32_t G, uint32_t xcds) {
    if (xcds <= 1 || G < 2 * xcds) return i;
    const uint32_t x = i % xcds, slot = i / xcds, q = G / xcds, r = G % xcds;
    return x * q + (x < r ? x : r) + slot;
}

__device__ __forceinline__ uint32_t wave_incl_scan32(uint32_t v) { return wave_incl_scan(v); }

// lower bound in a sorted LDS array: number of elements < v
template <typename T> __device__ __forceinline__ uint32_t lds_lower_bound(const T* arr, uint32_t n, T v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (arr[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// one residual gap at bit `rel` of the staged window (BVG:788-795): zeta_k from a 32-bit window when it fits, else the
// 64-bit decoders; returns the code length, 0 = does not fit 64 bits (fail over)
template <bool GEN> __device__ __forceinline__ uint32_t read_residual(const uint32_t* stage, uint32_t rel, bool zfast, uint32_t zk, int coding, uint64_t& val) {
    uint32_t len = 0; val = 0;
    if (zfast) {
        const uint32_t w = win32<LIN>(stage, rel);
        const uint32_t z = w ? (uint32_t)__builtin_clz(w) : 32u;
        const uint32_t nbz = z * zk + zk - 1, zt = z + 1 + nbz;
        if (zt < 32) {
            const uint32_t tt = (w << (z + 1)) >> (32u - nbz);
            const uint32_t leftv = 1u << (z * zk);
            if (tt < leftv) { val = tt + leftv - 1u; len = zt; }
            else { val = ((tt << 1) | ((w >> (31u - zt)) & 1u)) - 1u; len = zt + 1; }
        }
    }
    if (len == 0) {
        const uint64_t w = win64<LIN>(stage, rel);
        len = GEN ? decode_generic_w(w, coding, zk, &val) : zeta64(w, zk, val);
    }
    return len;
}

// zeta_k from a 32-bit window (k >= 2): returns the code length, 0 = the code does not fit 31 bits.  Straight-line (selects
// only), so several independent decodes interleave in one instruction stream.
__device__ __forceinline__ uint32_t zeta_fast32(uint32_t w, uint32_t zk, uint32_t& val) {
    const uint32_t z = w ? (uint32_t)__builtin_clz(w) : 32u;
    const uint32_t nbz = z * zk + zk - 1, zt = z + 1 + nbz;
    const uint32_t fits = zt < 32 ? 1u : 0u, fm = 0u - fits;                    // masks instead of ?: so the compiler keeps it straight-line
    const uint32_t nb = (nbz & fm) | (1u & ~fm), zz = z & fm;                    // every shift amount stays in range
    const uint32_t tt = (w << (zz + 1)) >> (32u - nb);
    const uint32_t leftv = 1u << (zz * zk);
    const uint32_t shortc = tt < leftv ? 1u : 0u, sm = 0u - shortc;
    const uint32_t ext = ((tt << 1) | ((w >> (31u - (zt & fm))) & 1u)) - 1u;
    val = ((tt + leftv - 1u) & sm) | (ext & ~sm);
    return (zt + 1u - shortc) & fm;
}

// Copy blocks in PREFIX form for the position tasks: entry i = (end position of block i in the referenced list) |
// (kept elements up to and including block i) << HS, so both directions of the copy mask (MaskedLongIterator.java:67-128) are
// binary searches instead of walks over up to hundreds of blocks: rank of a list position among the kept ones, and the
// position of the t-th kept element.  Blocks alternate keep / skip, block 0 keeps (and may be empty).
template <typename T> struct MaskPrefix {
    static constexpr uint32_t HS = sizeof(T) * 4;
    static __device__ __forceinline__ uint32_t pos(T e) { return (uint32_t)(e & (T)(((T)1 << HS) - 1)); }
    static __device__ __forceinline__ uint32_t kept(T e) { return (uint32_t)(e >> HS); }
    static __device__ __forceinline__ T pack(uint32_t p, uint32_t k) { return (T)p | (T)((T)k << HS); }
    // kept elements among list positions [0, qq); qn = the first kept position >= qq (rlen if none)
    static __device__ __forceinline__ uint32_t rank(const T* blk, uint32_t bc, uint32_t rlen, uint32_t qq, uint32_t& qn) {
        uint32_t lo = 0, hi = bc;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (pos(blk[mid]) > qq) hi = mid; else lo = mid + 1; }
        const T prev = lo ? blk[lo - 1] : (T)0;
        const bool keepb = lo < bc ? !(lo & 1u) : !(bc & 1u);                 // behind the last block: kept iff their number is even
        if (keepb) { qn = qq; return kept(prev) + (qq - pos(prev)); }
        qn = lo < bc ? pos(blk[lo]) : rlen;
        return kept(prev);
    }
    // the t-th kept element: its list position, the kept elements left in its block (kInf behind the last block), the next block
    static __device__ __forceinline__ void select(const T* blk, uint32_t bc, uint32_t rlen, uint32_t t, uint32_t& q, uint32_t& krem, uint32_t& bi) {
        uint32_t lo = 0, hi = bc;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (kept(blk[mid]) > t) hi = mid; else lo = mid + 1; }
        const T prev = lo ? blk[lo - 1] : (T)0;
        if (lo < bc) { q = pos(prev) + (t - kept(prev)); krem = kept(blk[lo]) - t; bi = lo + 1; }
        else if (!(bc & 1u)) { q = pos(prev) + (t - kept(prev)); krem = 0xFFFFFFFFu; bi = bc; }
        else { q = rlen; krem = 0xFFFFFFFFu; bi = bc; }
    }
    // a keep block ended at block index bi - 1: skip block bi, enter keep block bi + 1
    static __device__ __forceinline__ void next_block(const T* blk, uint32_t bc, uint32_t rlen, uint32_t& q, uint32_t& krem, uint32_t& bi) {
        if (bi >= bc) { q = rlen; krem = 0xFFFFFFFFu; return; }
        q = pos(blk[bi]); bi++;
        if (bi >= bc) krem = 0xFFFFFFFFu; else { krem = pos(blk[bi]) - q; bi++; }
    }
};

// checksum term of successor m (an id RELATIVE to the kernel's base: node_base, or the block base of a wide graph) of a node whose key is k1: k1 * m -- one
// v_mad_u64_u32 for 32-bit lists.  mix_keyed(k0, k1, m + base) = this + (k1 * base + k0), and the bracket is added once per node, times its outdegree
// (mix_node_const, bvg_device.h).
template <typename T> __device__ __forceinline__ uint64_t mix_node(uint32_t k1, T m) { return (uint64_t)k1 * (uint64_t)m; }

// ordering point for LDS traffic inside ONE wavefront (its lanes run in lock step and its LDS operations complete in
// order): only the compiler has to be kept from moving accesses across it
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace rows
}  // namespace bvg
