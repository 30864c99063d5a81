// bvg_kernels.hip — hand-written gfx950 kernels for BVGraph successor-list decoding.
//
// Replaces the reference's sequential hot loop (BVGraph.java:1164-1176 -> successors() :995-1097 ->
// Masked/Merged/Interval iterators) with a block-parallel formulation:
//
//   * the node range is cut into BLOCKS of ~equal compressed size (plan_boundaries); one wavefront
//     (a 64-thread workgroup) owns one block and walks it in ROWS of up to 64 consecutive nodes,
//     one node per lane, each lane entering its record through the offsets index;
//   * a row is decoded in two phases: (1) every lane parses its own record (outdegree, reference,
//     copy blocks, intervals, residual gaps -> absolute residuals) into LDS; (2) a lock-step
//     data-flow loop in which every lane emits ONE successor per iteration by a three-way merge of
//     {masked copy of the referenced list, intervals, residuals}; a lane whose referenced list is
//     being produced by a lower lane of the same row simply waits on that lane's `produced` counter,
//     so reference chains pipeline instead of serialising;
//   * the successor lists of the last `window` nodes stay in an LDS pool (compacted when full), so
//     a block can be arbitrarily long with a bounded LDS footprint; only the first nodes of a block
//     need a HALO: the (few) earlier nodes their reference chains reach, found by plan_halo;
//   * in scan mode successors are consumed on chip (count + checksum, one atomic per block); in
//     materialise mode each row's pool segment is copied out with coalesced 8-byte stores.
//
// Blocks that do not fit the LDS pool (a node whose list alone exceeds it) are handed to the same
// kernel instantiated over a global-memory pool (the slow path).
#include "bvg_kernels.h"

#include <algorithm>
#include <type_traits>

namespace bvg {

namespace {

constexpr uint32_t kInf = 0xFFFFFFFFu;


template <typename T> __device__ __forceinline__ T sentinel() { return (T)~(T)0; }

__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = __shfl_up(v, o, 64);
        if ((int)lane_id() >= o) v += t;
    }
    return v;
}

// GEN = false: the file uses BVGraph's default codings (gamma outdegrees / block counts / blocks, unary
// references, zeta_k residuals — BVGraph.java:527-542), decoded by straight-line code; GEN = true:
// any legal combination of compressionflags, dispatched per field (grid-uniform branches).
template <bool GEN> struct Rd {
    static __device__ __forceinline__ uint64_t outdegree(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.outdegree, 0, g); return c.read_gamma(g); }
    static __device__ __forceinline__ uint64_t reference(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.reference, 0, g); return c.read_unary(g); }
    static __device__ __forceinline__ uint64_t block_count(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.block_count, 0, g); return c.read_gamma(g); }
    static __device__ __forceinline__ uint64_t block(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.block, 0, g); return c.read_gamma(g); }
    static __device__ __forceinline__ uint64_t residual(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.residual, (unsigned)k.zeta_k, g); return c.read_zeta((unsigned)k.zeta_k, g); }
};

// RING: entries of the node ring — kRing for windows up to kMaxWindow; kRingBig (24 KiB of LDS, so its own instantiation: the
// giants of ordinary graphs run beside tier 0 and must not take its LDS) for the wide-window mode of the global-memory tier
template <typename T, bool MAT, bool SLOW, bool GEN, int RING = kRing>
__global__ void __launch_bounds__(64) decode_kernel(DecodeArgs a) {
    typedef Rd<GEN> R;
    constexpr uint32_t RM = RING - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];     // fast path: pool then scratch
    typedef typename std::conditional<SLOW, uint64_t, uint32_t>::type idx_t;
    __shared__ idx_t nd_base[RING];
    __shared__ uint32_t nd_d[RING];
    __shared__ uint32_t produced[64];
    __shared__ uint32_t scr_used;

    const unsigned lane = threadIdx.x;
    const uint32_t bid = a.work_list ? a.work_list[blockIdx.x] : (a.blk_lo + blockIdx.x);
    const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
    if (e <= a.from || s >= a.to || s >= e) return;
    const uint32_t halo = a.blk_halo[bid];
    const uint64_t hmask = a.blk_mask[bid];
    const int W = a.window;
    const int64_t rep_lo = s > a.from ? s : a.from, rep_hi = e < a.to ? e : a.to;

    T* const pool = SLOW ? reinterpret_cast<T*>(a.gpool) + (uint64_t)blockIdx.x * a.gpool_elems : reinterpret_cast<T*>(dyn_lds);
    T* const scr = SLOW ? reinterpret_cast<T*>(a.gscr) + (uint64_t)blockIdx.x * a.gscr_elems : reinterpret_cast<T*>(dyn_lds) + a.lds_pool_elems;
    const uint64_t CAP = SLOW ? a.gpool_elems : a.lds_pool_elems;
    const uint64_t SCR = SLOW ? a.gscr_elems : a.lds_scr_elems;
    // the stream window lives behind pool + scratch in the dynamic LDS block (slow path: at its start)
    uint32_t* const stage = reinterpret_cast<uint32_t*>(dyn_lds + (SLOW ? 0 : (size_t)(a.lds_pool_elems + a.lds_scr_elems) * sizeof(T)));
    const uint32_t stage_bytes = a.lds_stage_words * 4u;

    for (unsigned i = lane; i < (unsigned)RING; i += 64) { nd_base[i] = 0; nd_d[i] = 0; }
    __syncthreads();

    uint64_t pool_used = 0;
    uint64_t stg_bit0 = 0; uint32_t stg_bits = 0;             // staged window (wave-uniform)
    uint64_t blk_arcs = 0, blk_chk = 0, blk_nodes = 0;
    unsigned err = 0;
    bool failed = false;
    const bool halo_all = W > kMaxWindow;                     // wide windows: the halo is a count, every halo node is decoded

    int64_t r0 = s - (int64_t)halo;
    while (r0 < e) {
        // ------------------------------------------------------------------ row set-up
        const unsigned err_row0 = err;
        const int64_t x = r0 + lane;
        const bool in_range = x < e;
        const uint64_t hbit = x < s ? (uint64_t)(s - 1 - x) & 63u : 0;
        const bool needed = in_range && (x >= s || halo_all || ((hmask >> hbit) & 1ull));
        uint64_t off_x = 0, rec_end = 0;
        if (in_range) { off_x = a.offsets[x]; rec_end = a.offsets[x + 1]; }
        {   // (re)stage the LDS window when this row's records are not covered by it
            const int64_t left = e - r0;
            const uint64_t row_lo = __shfl(off_x, 0, 64);
            const uint64_t row_hi = __shfl(rec_end, left >= 64 ? 63 : (int)left - 1, 64);
            if (!(row_lo >= stg_bit0 && row_hi + 64 <= stg_bit0 + stg_bits)) {
                __syncthreads();
                const uint64_t b0 = (row_lo >> 3) & ~15ull;
                uint64_t nb = a.padded_bytes > b0 ? a.padded_bytes - b0 : 0;
                if (nb > stage_bytes) nb = stage_bytes;
                for (uint32_t c = lane; c < (uint32_t)(nb >> 4); c += 64) {
                    const uint4 v = *reinterpret_cast<const uint4*>(a.graph + b0 + ((uint64_t)c << 4));
                    uint4 w; w.x = __builtin_bswap32(v.x); w.y = __builtin_bswap32(v.y); w.z = __builtin_bswap32(v.z); w.w = __builtin_bswap32(v.w);
                    *reinterpret_cast<uint4*>(&stage[c << 2]) = w;
                }
                stg_bit0 = b0 << 3; stg_bits = (uint32_t)(nb << 3);
                __syncthreads();
            }
        }
        BitCursor cur{a.graph, off_x, a.limit_byte, stage, stg_bit0, stg_bits, 0u, 0xFFFFFFFFu};
        uint32_t d = 0;
        if (needed) d = (uint32_t)R::outdegree(cur, a.cod, rec_end);          // readOutdegree, BVG:654-660
        // how many leading lanes fit in the pool?
        uint64_t dclamp = d > CAP ? CAP + 1 : d;
        uint64_t incl = wave_incl_scan64(dclamp);
        uint64_t avail = CAP - pool_used;
        uint64_t total = __shfl(incl, 63, 64);
        if (total > avail && pool_used > 0) {
            // compact: keep only the lists of the last W nodes, moved to the front of the pool
            uint64_t packed = 0;                                              // lists already moved to the front
            for (int c0 = 0; c0 < W; c0 += 64) {                              // 64 of the W window nodes at a time, oldest first
                uint64_t my_d = 0, my_base = 0; const int64_t y = r0 - W + c0 + (int64_t)lane;
                const bool livelane = c0 + (int)lane < W && y >= s - (int64_t)halo && y >= 0;
                if (livelane) { my_d = nd_d[(uint64_t)y & RM]; my_base = nd_base[(uint64_t)y & RM]; }
                const uint64_t nincl = wave_incl_scan64(my_d);
                const uint64_t nbase = packed + nincl - my_d;
                const int cn = W - c0 < 64 ? W - c0 : 64;
                for (int j = 0; j < cn; j++) {
                    const uint64_t src = __shfl(my_base, j, 64), dst = __shfl(nbase, j, 64), len = __shfl(my_d, j, 64);
                    if (src != dst)
                        for (uint64_t t0 = 0; t0 < len; t0 += 64) {                                              // (lists lie in node order: a move never lands on one not yet moved;
                            const uint64_t t = t0 + lane; T v = 0;                                               //  every element of a step is read before any is written)
                            if (t < len) v = pool[src + t];
                            __syncthreads();
                            if (t < len) pool[dst + t] = v;
                            __syncthreads();
                        }
                }
                if (livelane) nd_base[(uint64_t)y & RM] = (idx_t)nbase;
                packed += __shfl(nincl, 63, 64);
            }
            pool_used = packed;
            avail = CAP - pool_used;
            __syncthreads();
        }
        unsigned k = 64;
        if (total > avail) k = (unsigned)__popcll(ballot(incl <= avail));   // incl is monotone: a prefix of lanes
        {
            int64_t left_in_block = e - r0;
            if ((int64_t)k > left_in_block) k = (unsigned)left_in_block;
        }
        if (k == 0) { failed = true; break; }                                // first node alone overflows the pool
        const bool act = needed && lane < k;
        const uint64_t base = pool_used + (incl - dclamp);
        if (act) { nd_base[(uint64_t)x & RM] = (idx_t)base; nd_d[(uint64_t)x & RM] = d; }
        pool_used += __shfl(incl, (int)k - 1, 64);
        if (lane == 0) scr_used = 0;
        produced[lane] = act ? 0u : kInf;
        __syncthreads();

        // ------------------------------------------------------------------ phase 1: parse own record
        uint32_t ref = 0, bc = 0, ic = 0, nres = 0;
        uint64_t sb = 0, ib = 0;
        bool overflow = false;
        if (act && d > 0) {
            if (W > 0) {                                                     // BVG:1015
                uint64_t r = R::reference(cur, a.cod, rec_end);             // readReference, BVG:692-703
                if (r > (uint64_t)W || (int64_t)r > x) { err |= ERR_REF_RANGE; r = 0; }
                ref = (uint32_t)r;
            }
            int64_t extra = d;
            if (ref > 0) {                                                   // BVG:1020-1032
                uint64_t nb = R::block_count(cur, a.cod, rec_end);
                if (nb > rec_end - (cur.pos < rec_end ? cur.pos : rec_end) + 1) { err |= ERR_OVERRUN; nb = 0; }
                bc = (uint32_t)nb;
                sb = atomicAdd(&scr_used, bc);
                if (sb + bc > SCR) { overflow = true; bc = 0; }
                int64_t copied = 0, tot = 0;
                for (uint32_t i = 0; i < bc; i++) {
                    uint64_t b = R::block(cur, a.cod, rec_end) + (i ? 1 : 0);
                    scr[sb + i] = (T)b;
                    tot += (int64_t)b;
                    if (!(i & 1)) copied += (int64_t)b;
                    if (cur.pos > rec_end) { err |= ERR_OVERRUN; bc = i + 1; break; }
                }
                if (!(bc & 1)) copied += (int64_t)nd_d[(uint64_t)(x - ref) & RM] - tot;   // BVG:1030
                extra = (int64_t)d - copied;
                if (extra < 0 || copied < 0) { err |= ERR_MALFORMED; extra = 0; }      // never let a tail start before the list
            }
            if (extra > 0 && a.min_interval != 0) {                          // BVG:1037-1060 (always gamma)
                uint64_t ni = cur.read_gamma(rec_end);
                if (ni > (rec_end - (cur.pos < rec_end ? cur.pos : rec_end)) / 2 + 1) { err |= ERR_OVERRUN; ni = 0; }
                ic = (uint32_t)ni;
                ib = atomicAdd(&scr_used, 2 * ic);
                if (ib + 2ull * ic > SCR) { overflow = true; ic = 0; }
                int64_t prev = 0;
                for (uint32_t i = 0; i < ic; i++) {
                    int64_t left = i == 0 ? x + nat2int(cur.read_gamma(rec_end)) : prev + 1 + (int64_t)cur.read_gamma(rec_end);
                    int64_t len = (int64_t)cur.read_gamma(rec_end) + a.min_interval;
                    prev = left + len;
                    extra -= len;
                    scr[ib + 2 * i] = (T)left; scr[ib + 2 * i + 1] = (T)len;
                    if (cur.pos > rec_end) { err |= ERR_OVERRUN; ic = i + 1; break; }
                }
                if (extra < 0) { err |= ERR_MALFORMED; extra = 0; }
            }
            nres = (uint32_t)extra;
            if (nres > 0 && !overflow) {                                     // ResidualLongIterator, BVG:902-935
                T* tail = pool + base + d - nres;
                int64_t r = x + nat2int(R::residual(cur, a.cod, rec_end));
                tail[0] = (T)r;
                for (uint32_t t = 1; t < nres; t++) {
                    r += (int64_t)R::residual(cur, a.cod, rec_end) + 1;
                    tail[t] = (T)r;
                    if (cur.pos > rec_end) { err |= ERR_OVERRUN; break; }
                }
            }
            if (cur.pos != rec_end && !overflow) err |= ERR_MALFORMED;       // SURVEY A.6 self-check
        }
        if (ballot(overflow)) { failed = true; break; }
        __syncthreads();

        // ------------------------------------------------------------------ phase 2: data-flow emission
        const bool rep = act && x >= rep_lo && x < rep_hi;
        uint32_t k0 = 0, k1 = 0;
        if (rep && !MAT) node_key((uint64_t)x + a.node_base, k0, k1);
        T* const out = pool + base;
        const T* rl = pool; uint32_t rlen = 0, rpos = 0, keep = 0, bi = 0; int rlane = -1;
        if (act && ref > 0) {
            const int64_t y = x - ref;
            rl = pool + nd_base[(uint64_t)y & RM]; rlen = nd_d[(uint64_t)y & RM];
            if (y >= r0) rlane = (int)lane - (int)ref;
            if (bc == 0) keep = kInf;                                        // MaskedLongIterator.java:73-78
            else {
                keep = (uint32_t)scr[sb]; bi = 1;
                if (keep == 0) {
                    if (bi >= bc) rpos = rlen;
                    else { rpos += (uint32_t)scr[sb + bi]; bi++; if (bi >= bc) keep = kInf; else { keep = (uint32_t)scr[sb + bi]; bi++; } }
                }
            }
        }
        T ivcur = 0; uint32_t ivrem = 0, ivi = 0;
        if (ic > 0) { ivcur = scr[ib]; ivrem = (uint32_t)scr[ib + 1]; ivi = 1; }
        uint32_t rsi = 0;
        T rhead = nres ? out[d - nres] : sentinel<T>();
        uint32_t j = 0;
        uint64_t chk = 0;
        for (;;) {
            const bool todo = act && j < d;
            if (!ballot(todo)) break;
            const bool cneed = todo && rpos < rlen;
            const bool cready = !cneed || rlane < 0 || __hip_atomic_load(&produced[rlane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) > rpos;
            if (todo && cready) {
                const T c = cneed ? rl[rpos] : sentinel<T>();
                const T iv = ivrem ? ivcur : sentinel<T>();
                T m = c < iv ? c : iv; m = m < rhead ? m : rhead;            // MergedLongIterator.java:63-92, three-way
                out[j] = m;
                j++;
                if (!MAT && rep) {
                    const uint64_t y64 = m == sentinel<T>() ? ~0ull : (uint64_t)m + a.node_base;
                    chk += mix_keyed(k0, k1, y64);
                }
                if (cneed && c == m) {                                       // MaskedLongIterator.java:81-100
                    rpos++;
                    if (--keep == 0) {
                        if (bi >= bc) rpos = rlen;
                        else { rpos += (uint32_t)scr[sb + bi]; bi++; if (bi >= bc) keep = kInf; else { keep = (uint32_t)scr[sb + bi]; bi++; } }
                    }
                }
                if (ivrem && iv == m) {                                      // LongIntervalSequenceIterator.java:71-78
                    ivcur++;
                    if (--ivrem == 0 && ivi < ic) { ivcur = scr[ib + 2 * ivi]; ivrem = (uint32_t)scr[ib + 2 * ivi + 1]; ivi++; }
                }
                if (rsi < nres && rhead == m) { rsi++; rhead = rsi < nres ? out[d - nres + rsi] : sentinel<T>(); }
                __hip_atomic_store(&produced[lane], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        if (rep) { blk_arcs += d; blk_chk += chk; blk_nodes += 1; }

        // ------------------------------------------------------------------ materialise: coalesced copy-out
        if (MAT) {
            __syncthreads();
            const uint64_t repmask = ballot(rep);
            if (repmask) {
                const int la = __ffsll((unsigned long long)repmask) - 1;
                const int lb = 63 - __clzll(repmask);
                const uint64_t seg0 = __shfl(base, la, 64);
                const uint64_t seg1 = __shfl(base + d, lb, 64);
                const uint64_t dst0 = a.batch ? a.cum[bid >> 1] : a.cum[(r0 + la) - a.from];
                for (uint64_t t = lane; t < seg1 - seg0; t += 64) {
                    const T v = pool[seg0 + t];
                    a.succ[dst0 + t] = v == sentinel<T>() ? -1ll : (int64_t)((uint64_t)v + a.node_base);
                }
                if (rep && a.outdeg && !a.batch) a.outdeg[x - a.from] = (int32_t)d;
            }
        }
        // a wide-window halo is decoded whole: its nodes that lie on no chain of this block may reference lists from before the
        // halo, which were never decoded; whatever they report is void (each is checked as a node of its own block)
        if (halo_all && x < s) err = err_row0;
        __syncthreads();
        r0 += k;
    }

    err = wave_or32(err);
    if (failed) {
        if (lane == 0) {
            uint32_t slot = atomicAdd(a.fail_count, 1u);
            if (slot < a.fail_cap) a.fail_list[slot] = bid;
        }
        return;
    }
    blk_arcs = wave_sum64(blk_arcs); blk_chk = wave_sum64(blk_chk); blk_nodes = wave_sum64(blk_nodes);
    if (lane == 0) {
        unsigned long long* const accs = a.acc + (size_t)(bid & a.acc_mask) * kAccStride;   // this block's result stripe
        atomicAdd(&accs[0], (unsigned long long)blk_arcs);
        atomicAdd(&accs[1], (unsigned long long)blk_chk);
        atomicAdd(&accs[2], (unsigned long long)blk_nodes);
        if (err) atomicOr(&accs[3], (unsigned long long)err);
    }
}

// ---------------------------------------------------------------------------------------------
__global__ void outdegree_kernel(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, int64_t from, int64_t to,
                                 int coding, int32_t* out, unsigned long long* total) {
    int64_t x = from + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t d = 0;
    if (x < to) {
        BitCursor cur{graph, offsets[x], limit_byte};
        d = cur.read_coded(coding, 0, offsets[x + 1]);
        out[x - from] = (int32_t)d;
    }
    if (total) {
        d = wave_sum64(d);
        if ((threadIdx.x & 63) == 0 && d) atomicAdd(total, (unsigned long long)d);
    }
}

__global__ void outdegree_gather_kernel(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, const int64_t* nodes, int64_t count,
                                        int coding, int32_t* out, uint64_t* first) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t x = nodes[i];
    BitCursor cur{graph, offsets[x], limit_byte};
    out[i] = (int32_t)cur.read_coded(coding, 0, offsets[x + 1]);
    first[2 * i] = (uint64_t)x; first[2 * i + 1] = (uint64_t)x + 1;
}

// ---- exclusive scan int32 -> uint64, three phases, 1024 elements per workgroup ----
constexpr int kScanTile = 1024;
__global__ void scan_partials(const int32_t* in, int64_t n, uint64_t* partial) {
    __shared__ uint64_t wsum[4];
    int64_t i0 = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 4;
    uint64_t v = 0;
    for (int t = 0; t < 4; t++) if (i0 + t < n) v += (uint64_t)(uint32_t)in[i0 + t];
    v = wave_sum64(v);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ void scan_partials_serial(uint64_t* partial, int64_t nparts) {
    // one wavefront: 64-wide chunks with a running carry
    uint64_t carry = 0;
    for (int64_t i0 = 0; i0 < nparts; i0 += 64) {
        int64_t i = i0 + threadIdx.x;
        uint64_t v = i < nparts ? partial[i] : 0;
        uint64_t inc = wave_incl_scan64(v);
        if (i < nparts) partial[i] = carry + inc - v;
        carry += __shfl(inc, 63, 64);
    }
}
__global__ void scan_final(const int32_t* in, int64_t n, const uint64_t* partial, uint64_t* out) {
    __shared__ uint64_t wsum[4];
    int64_t i0 = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 4;
    uint64_t e[4]; uint64_t v = 0;
    for (int t = 0; t < 4; t++) { e[t] = (i0 + t < n) ? (uint64_t)(uint32_t)in[i0 + t] : 0; v += e[t]; }
    uint64_t inc = wave_incl_scan64(v);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint64_t off = partial[blockIdx.x];
    for (unsigned w = 0; w < (threadIdx.x >> 6); w++) off += wsum[w];
    uint64_t run = off + inc - v;
    for (int t = 0; t < 4; t++) { if (i0 + t < n) out[i0 + t] = run; run += e[t]; }
    if (i0 <= n - 1 && n - 1 < i0 + 4) out[n] = run;     // total at out[n]
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

// ---- plan ----
__global__ void plan_boundaries_kernel(Offsets offsets, int64_t n, uint64_t block_bits, uint64_t nb, uint64_t* first) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nb) return;
    if (j == nb) { first[j] = (uint64_t)n; return; }
    // first node whose record starts at or after bit j*block_bits
    const uint64_t target = j * block_bits;
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t mid = lo + ((hi - lo) >> 1); if (offsets[mid] < target) lo = mid + 1; else hi = mid; }
    first[j] = (uint64_t)lo;
}

__global__ void plan_halo_kernel(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, int64_t n, const uint64_t* first,
                                 uint32_t nblk, int window, Codings cod, uint32_t* halo, uint64_t* mask) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nblk) return;
    const int64_t s = (int64_t)first[k];
    const bool wide = window > kMaxWindow;                                    // halo as a count (up to kMaxHaloBig), no mask
    int64_t reach = 0;
    uint64_t m = 0; bool bad = false;
    if (window > 0 && s > 0) {
        const int64_t xe = s + window < n ? s + window : n;
        for (int64_t x = s; x < xe && !bad; x++) {
            int64_t y = x;
            for (;;) {                                                       // follow the reference chain of x
                BitCursor cur{graph, offsets[y], limit_byte};
                const uint64_t end = offsets[y + 1];
                uint64_t d = cur.read_coded(cod.outdegree, 0, end);
                if (d == 0) break;
                uint64_t r = cur.read_coded(cod.reference, 0, end);
                if (r == 0 || r > (uint64_t)window || (int64_t)r > y) break;
                y -= (int64_t)r;
                if (y < s) {
                    int64_t dist = s - 1 - y;
                    if (dist >= (wide ? kMaxHaloBig : kMaxHalo)) { bad = true; break; }
                    if (wide) reach = dist + 1 > reach ? dist + 1 : reach; else m |= 1ull << dist;
                }
            }
        }
    }
    if (wide) { halo[k] = bad ? 0xFFFFFFFFu : (uint32_t)reach; mask[k] = ~0ull; return; }
    halo[k] = bad ? 0xFFFFFFFFu : (m ? 64u - (uint32_t)__builtin_clzll(m) : 0u);
    mask[k] = m;
}

// one wavefront per block: lanes stride over the block's nodes (coalesced offsets reads); the value kept is
// the largest "own list + the W lists before it" — what the LDS pool must hold to decode that node at all
__global__ void plan_maxd_kernel(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, const uint64_t* first, const uint32_t* halo,
                                 uint32_t nblk, int coding, int window, uint32_t* maxd, uint64_t* bign, uint32_t* bigd) {
    const uint32_t k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= nblk) return;
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t h = halo[k] == 0xFFFFFFFFu ? 0u : halo[k];
    const int64_t lo = (int64_t)first[k] - (int64_t)h, hi = (int64_t)first[k + 1];
    uint64_t m = 0, mrec = 0;
    uint32_t bd = 0; uint64_t bn = first[k];                                  // the block's own longest list and its node (not the halo's)
    uint32_t prevd = 0;                                                       // d of node x-64 (previous chunk, same lane)
    for (int64_t x0 = lo; x0 < hi; x0 += 64) {
        const int64_t x = x0 + lane;
        uint32_t dd = 0;
        if (x < hi) {
            BitCursor cur{graph, offsets[x], limit_byte};
            const uint64_t end = offsets[x + 1];
            const uint64_t d = cur.read_coded(coding, 0, end);
            dd = d > 0x3FFFFFFFull ? 0x3FFFFFFFu : (uint32_t)d;
            const uint64_t rec = end - offsets[x];
            mrec = rec > mrec ? rec : mrec;
            if (x >= (int64_t)first[k] && dd > bd) { bd = dd; bn = (uint64_t)x; }
        }
        uint64_t need = dd;
        for (int j = 1; j <= window && j < 64; j++) {
            const uint32_t a = __shfl_up(dd, j, 64);                           // same chunk
            const uint32_t b = __shfl(prevd, (int)((lane + 64 - j) & 63), 64); // previous chunk
            need += (int)lane >= j ? a : b;
        }
        if (x < hi) m = need > m ? need : m;
        prevd = dd;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t t = __shfl_xor(m, o, 64); m = t > m ? t : m; const uint64_t t2 = __shfl_xor(mrec, o, 64); mrec = t2 > mrec ? t2 : mrec;
        const uint32_t d2 = __shfl_xor(bd, o, 64); const uint64_t n2 = __shfl_xor(bn, o, 64);
        if (d2 > bd || (d2 == bd && n2 < bn)) { bd = d2; bn = n2; }
    }
    if (lane == 0 && bign) { bign[k] = bn; bigd[k] = bd; }
    // bit 31 flags a record longer than the LDS stream window (4 KiB): such a block goes straight to the global tier.  (Tried: a
    // 16 KiB window in the largest LDS class for records up to 128 Kbit — 78 KiB workgroups beside tier 0 instead of one-wavefront
    // blocks with a 4 KiB footprint cost the 8 GiB eu scan 2.4 %, whether the plan filed the long records there or not.)
    if (lane == 0) maxd[k] = (m > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)m) | (mrec + 128 > 32768 ? 0x80000000u : 0u);
}

// ---- synthetic tiling ----
__device__ __forceinline__ uint64_t load_bits64(const uint8_t* src, uint64_t bitpos) {
    const uint8_t* p = src + (bitpos >> 3);
    uint64_t hi = __builtin_bswap64(*reinterpret_cast<const u64*>(p));
    unsigned sh = (unsigned)bitpos & 7u;
    uint64_t w = hi << sh;
    if (sh) w |= (uint64_t)p[8] >> (8u - sh);
    return w;
}
// A MOSAIC: the streams of up to kMosaicMax base graphs concatenated, the whole cycle repeated `cycles` times (BV records are
// translation invariant: a record decodes to the same lists shifted by the node id it is given).  One base = the old bvg_tile.
__global__ void mosaic_graph_kernel(MosaicSrc m, uint8_t* dst, uint64_t dst_words, uint64_t total_bits) {
    // grid-stride: a launch may not exceed 2^32 threads (a 47 GB stream has 5.9e9 words)
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < dst_words; w += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t P = w * 64;
        uint64_t val = 0;
        if (P < total_bits) {
            uint64_t off = P % m.cycle_bits;
            int k = 0;
            while (k + 1 < m.k && off >= m.bit_prefix[k + 1]) k++;
            off -= m.bit_prefix[k];
            unsigned got = 0;
            while (got < 64) {                                                // a word may straddle several (short) sources
                const uint64_t rem = m.bits[k] - off;
                uint64_t piece = load_bits64(m.graph[k], off);
                const uint64_t take = rem < 64u - got ? rem : 64u - got;
                piece &= take >= 64 ? ~0ull : ~(~0ull >> take);
                val |= piece >> got;
                got += (unsigned)take;
                off += take;
                if (off >= m.bits[k]) { off = 0; k = k + 1 < m.k ? k + 1 : 0; }
            }
            if (total_bits - P < 64) val &= ~(~0ull >> (total_bits - P));
        }
        reinterpret_cast<uint64_t*>(dst)[w] = __builtin_bswap64(val);
    }
}
__device__ __forceinline__ uint64_t mosaic_offset(const MosaicSrc& m, int64_t tot, int64_t i) {
    if (i >= tot) return (uint64_t)(tot / m.cycle_nodes) * m.cycle_bits;
    const int64_t c = i / m.cycle_nodes; int64_t r = i - c * m.cycle_nodes;
    int k = 0;
    while (k + 1 < m.k && r >= m.node_prefix[k + 1]) k++;
    return (uint64_t)c * m.cycle_bits + m.bit_prefix[k] + m.offs[k][r - m.node_prefix[k]];
}
// offsets of the mosaic, written in packed form straight away; *overflow != 0 if 2^kOffShift consecutive records span 2^32 bits or more
__global__ void mosaic_offsets_kernel(MosaicSrc m, int64_t tot, uint32_t* dst_lo, uint64_t* dst_hi, unsigned* overflow) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= tot; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t v = mosaic_offset(m, tot, i);
        const uint64_t b = mosaic_offset(m, tot, i & ~(((int64_t)1 << kOffShift) - 1));
        if (v - b > 0xFFFFFFFFull) atomicOr(overflow, 1u);
        dst_lo[i] = (uint32_t)(v - b);
        if ((i & (((int64_t)1 << kOffShift) - 1)) == 0) dst_hi[i >> kOffShift] = v;
    }
}

// successors as 32-bit ids for the host path of graphs with at most 2^32 nodes (half the bytes over PCIe): -1 (a short list of a
// malformed stream) becomes 0xFFFFFFFF
__global__ void narrow_succ_kernel(const int64_t* in, uint32_t* out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)in[i];
}

__global__ void pack_offsets_kernel(const uint64_t* src, int64_t first, int64_t count, uint32_t* lo, uint64_t* hi, unsigned* overflow) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = first + j;
        const uint64_t v = src[j], b = src[j & ~(((int64_t)1 << kOffShift) - 1)];
        if (v < b || v - b > 0xFFFFFFFFull) { atomicOr(overflow, 1u); continue; }
        lo[x] = (uint32_t)(v - b);
        if ((x & (((int64_t)1 << kOffShift) - 1)) == 0) hi[x >> kOffShift] = v;
    }
}

__global__ void unpack_offsets_kernel(Offsets o, int64_t first, int64_t count, uint64_t* dst) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (int64_t)gridDim.x * blockDim.x) dst[j] = o[first + j];
}

}  // namespace

// ------------------------------------------------------------------------------------------------
void launch_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, bool materialise, bool slow, hipStream_t s) {
    if (nblocks == 0) return;
    dim3 grid(nblocks), block(64);
    const bool gen = !(a.cod.outdegree == BVG_GAMMA && a.cod.reference == BVG_UNARY && a.cod.block_count == BVG_GAMMA &&
                       a.cod.block == BVG_GAMMA && a.cod.residual == BVG_ZETA);
    const size_t dyn = (slow ? 0 : (size_t)(a.lds_pool_elems + a.lds_scr_elems) * (wide ? 8 : 4)) + (size_t)a.lds_stage_words * 4;
    if (a.window > kMaxWindow) {                                              // wide windows: global-memory tier only, big node ring
        if (!slow) return;
        if (!wide) { if (!materialise) hipLaunchKernelGGL((decode_kernel<uint32_t, false, true, true, kRingBig>), grid, block, dyn, s, a);
                     else hipLaunchKernelGGL((decode_kernel<uint32_t, true, true, true, kRingBig>), grid, block, dyn, s, a); }
        else { if (!materialise) hipLaunchKernelGGL((decode_kernel<uint64_t, false, true, true, kRingBig>), grid, block, dyn, s, a);
               else hipLaunchKernelGGL((decode_kernel<uint64_t, true, true, true, kRingBig>), grid, block, dyn, s, a); }
        return;
    }
#define BVG_LAUNCH2(T, M, S) do { if (gen) hipLaunchKernelGGL((decode_kernel<T, M, S, true>), grid, block, dyn, s, a); \
                                  else hipLaunchKernelGGL((decode_kernel<T, M, S, false>), grid, block, dyn, s, a); } while (0)
#define BVG_LAUNCH(T, M, S) BVG_LAUNCH2(T, M, S)
    if (!wide) {
        if (!materialise) { if (!slow) BVG_LAUNCH(uint32_t, false, false); else BVG_LAUNCH(uint32_t, false, true); }
        else { if (!slow) BVG_LAUNCH(uint32_t, true, false); else BVG_LAUNCH(uint32_t, true, true); }
    } else {
        if (!materialise) { if (!slow) BVG_LAUNCH(uint64_t, false, false); else BVG_LAUNCH(uint64_t, false, true); }
        else { if (!slow) BVG_LAUNCH(uint64_t, true, false); else BVG_LAUNCH(uint64_t, true, true); }
    }
#undef BVG_LAUNCH2
#undef BVG_LAUNCH
}

void launch_outdegrees(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, int64_t from, int64_t to, int coding,
                       int32_t* out, unsigned long long* total, hipStream_t s) {
    int64_t n = to - from;
    if (n <= 0) return;
    hipLaunchKernelGGL(outdegree_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, graph, limit_byte, offsets, from, to, coding, out, total);
}

void launch_outdegrees_gather(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, const int64_t* nodes, int64_t count,
                              int coding, int32_t* out, uint64_t* first, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(outdegree_gather_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, graph, limit_byte, offsets, nodes, count, coding, out, first);
}

size_t scan_tmp_elems(int64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile) + 1; }

void launch_exclusive_scan(const int32_t* in, uint64_t* out, int64_t n, uint64_t* tmp, hipStream_t s) {
    int64_t parts = (n + kScanTile - 1) / kScanTile;
    if (parts == 0) parts = 1;
    hipLaunchKernelGGL(scan_partials, dim3((unsigned)parts), dim3(256), 0, s, in, n, tmp);
    hipLaunchKernelGGL(scan_partials_serial, dim3(1), dim3(64), 0, s, tmp, parts);
    hipLaunchKernelGGL(scan_final, dim3((unsigned)parts), dim3(256), 0, s, in, n, tmp, out);
}

// result stripes -> stripe 0: sums of the three counters, OR of the error bits
__global__ void __launch_bounds__(256) reduce_acc_kernel(unsigned long long* acc, uint32_t stripes) {
    __shared__ unsigned long long part[4][4];
    unsigned long long v[4] = {0, 0, 0, 0};
    for (uint32_t s = threadIdx.x; s < stripes; s += 256) {
        const unsigned long long* p = acc + (size_t)s * kAccStride;
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] |= p[3];
    }
    v[0] = wave_sum64(v[0]); v[1] = wave_sum64(v[1]); v[2] = wave_sum64(v[2]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[3] |= __shfl_xor(v[3], o, 64);
    if ((threadIdx.x & 63u) == 0) for (int i = 0; i < 4; i++) part[threadIdx.x >> 6][i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        acc[0] = part[0][0] + part[1][0] + part[2][0] + part[3][0];
        acc[1] = part[0][1] + part[1][1] + part[2][1] + part[3][1];
        acc[2] = part[0][2] + part[1][2] + part[2][2] + part[3][2];
        acc[3] = part[0][3] | part[1][3] | part[2][3] | part[3][3];
    }
}
// 64-bit hash of a device array: the sum, mod 2^64, of a position-keyed mix of its 8-byte words (the tail bytes packed into one more
// word).  Order-free, so a grid-stride pass at memory speed; ties the index on disk (bvg_save_index) to EVERY byte of the stream it was
// built from, and guards the index payload itself against bit rot.
__global__ void __launch_bounds__(256) hash_words_kernel(const uint8_t* p, uint64_t nbytes, unsigned long long* out) {
    const uint64_t nw = nbytes >> 3;
    const uint64_t* w = reinterpret_cast<const uint64_t*>(p);
    uint64_t h = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t z = w[i] + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
        h += z;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (nbytes & 7)) {
        uint64_t t = 0;
        for (uint64_t b = nw << 3; b < nbytes; b++) t = (t << 8) | p[b];
        uint64_t z = t + (nw + 1) * 0x9E3779B97F4A7C15ull + (nbytes & 7);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
        h += z;
    }
    h = wave_sum64(h);
    if ((threadIdx.x & 63u) == 0 && h) atomicAdd(out, (unsigned long long)h);
}
void launch_hash_words(const void* p, uint64_t nbytes, unsigned long long* out, hipStream_t s) {
    const uint64_t nw = nbytes >> 3;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((nw + 255) / 256, 1), 256u * 32u);
    hipLaunchKernelGGL(hash_words_kernel, dim3(blocks), dim3(256), 0, s, (const uint8_t*)p, nbytes, out);
}

void launch_reduce_acc(unsigned long long* acc, uint32_t stripes, hipStream_t s) {
    hipLaunchKernelGGL(reduce_acc_kernel, dim3(1), dim3(256), 0, s, acc, stripes);
}

void launch_plan_boundaries(Offsets offsets, int64_t n, uint64_t block_bits, uint64_t nb, uint64_t* first, hipStream_t s) {
    hipLaunchKernelGGL(plan_boundaries_kernel, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0, s, offsets, n, block_bits, nb, first);
}

void launch_plan_halo(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, int64_t n, const uint64_t* first, uint32_t nblk,
                      int window, Codings cod, uint32_t* halo, uint64_t* mask, hipStream_t s) {
    if (!nblk) return;
    hipLaunchKernelGGL(plan_halo_kernel, dim3((nblk + 127) / 128), dim3(128), 0, s, graph, limit_byte, offsets, n, first, nblk, window, cod, halo, mask);
}

// the longest record of every block: its node and its length in bits (one wavefront per block)
__global__ void plan_longest_kernel(Offsets offsets, const uint64_t* first, uint32_t nblk, uint64_t* node, uint64_t* bits) {
    const uint32_t k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= nblk) return;
    const unsigned lane = threadIdx.x & 63u;
    uint64_t best = 0, at = first[k];
    for (uint64_t x = first[k] + lane; x < first[k + 1]; x += 64) {
        const uint64_t len = offsets[(int64_t)x + 1] - offsets[(int64_t)x];
        if (len > best) { best = len; at = x; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t b2 = __shfl_xor(best, o, 64), a2 = __shfl_xor(at, o, 64);
        if (b2 > best || (b2 == best && a2 < at)) { best = b2; at = a2; }
    }
    if (lane == 0) { node[k] = at; bits[k] = best; }
}

void launch_plan_longest(Offsets offsets, const uint64_t* first, uint32_t nblk, uint64_t* node, uint64_t* bits, hipStream_t s) {
    if (!nblk) return;
    hipLaunchKernelGGL(plan_longest_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, offsets, first, nblk, node, bits);
}

void launch_plan_maxd(const uint8_t* graph, uint64_t limit_byte, Offsets offsets, const uint64_t* first, const uint32_t* halo, uint32_t nblk,
                      int coding, int window, uint32_t* maxd, uint64_t* bign, uint32_t* bigd, hipStream_t s) {
    if (!nblk) return;
    hipLaunchKernelGGL(plan_maxd_kernel, dim3((nblk + 3) / 4), dim3(256), 0, s, graph, limit_byte, offsets, first, halo, nblk, coding, window, maxd, bign, bigd);
}

void launch_mosaic_graph(const MosaicSrc& m, uint8_t* dst, uint64_t dst_bytes, int64_t cycles, hipStream_t s) {
    const uint64_t words = dst_bytes / 8;
    hipLaunchKernelGGL(mosaic_graph_kernel, dim3((unsigned)std::min<uint64_t>((words + 255) / 256, 1u << 22)), dim3(256), 0, s, m, dst, words, m.cycle_bits * (uint64_t)cycles);
}
void launch_mosaic_offsets(const MosaicSrc& m, int64_t cycles, uint32_t* dst_lo, uint64_t* dst_hi, unsigned* overflow, hipStream_t s) {
    const int64_t tot = m.cycle_nodes * cycles;
    hipLaunchKernelGGL(mosaic_offsets_kernel, dim3((unsigned)std::min<int64_t>((tot + 256) / 256, 1 << 22)), dim3(256), 0, s, m, tot, dst_lo, dst_hi, overflow);
}
void launch_narrow_succ(const int64_t* in, uint32_t* out, uint64_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(narrow_succ_kernel, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 1u << 20)), dim3(256), 0, s, in, out, n);
}
void launch_pack_offsets(const uint64_t* src, int64_t first, int64_t count, uint32_t* lo, uint64_t* hi, unsigned* overflow, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(pack_offsets_kernel, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 1 << 20)), dim3(256), 0, s, src, first, count, lo, hi, overflow);
}
void launch_unpack_offsets(Offsets o, int64_t first, int64_t count, uint64_t* dst, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(unpack_offsets_kernel, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 1 << 20)), dim3(256), 0, s, o, first, count, dst);
}

}  // namespace bvg
