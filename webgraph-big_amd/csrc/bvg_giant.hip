// bvg_giant.hip — tier 2a: blocks holding a list (or a record) too large for the LDS row kernels.
//
// Transposed and social graphs have lists of 10^5..10^7 successors.  The generic global-memory kernel (bvg_kernels.hip,
// decode_kernel<SLOW>) follows the reference's iterators literally with ONE lane per list: ~1 us per successor, so a handful of
// such lists takes longer than the rest of the graph.  Here one workgroup of GNT = 512 threads walks its block node by node and
// every node is decoded by the whole workgroup, everything in a per-workgroup area of global memory:
//   * the counts of the record header (BVG:1003-1021, 1040) are decoded by wavefront 0 in step, on the scalar unit, from a register
//     bit buffer over a sliding LDS window of the stream;
//   * the copy blocks (BVG:1023-1032, kept in prefix form), the intervals (BVG:1042-1058, kept as {left, elements before}) and the
//     residuals (ResidualLongIterator, BVG:902-935) are cut at the entries of the skip index -- every kSkipEvery-th code of a long
//     section: its bit offset and the running sums before it -- into tasks of <= kSkipEvery codes, one per thread, read straight
//     from the stream in global memory;
//   * where the index has no entries yet (first scan, index build, BVG_NOSKIP) wavefront 0 resolves the code boundaries itself: lane
//     b decodes the code that would start at bit b of the next 128 bits, the chain of the codes that really start there is followed
//     with v_readlane, and prefix scans over its lanes turn up to 64 gaps per step into values (sections under 48 codes: in step);
//   * the list is put together by output POSITION as in the row kernels (bvg_rows.hip): every extra (interval, residual) finds
//     its place by binary searches (extras below it + copied elements below it: lower bound in the referenced list, rank under
//     the copy mask), then GNT equal tasks of consecutive positions fill in the kept elements of the referenced list
//     (MaskedLongIterator.java:73-100) and the interval elements (LongIntervalSequenceIterator.java:71-78).
// Streams whose three parts overlap (MergedLongIterator.java:85-89 would emit the value once), counts that contradict each other,
// non-default codings and windows > 64 fail over to decode_kernel<SLOW>; a work area that is too small is reported as such and
// the host retries with a larger one.
#include "bvg_rows_common.h"

namespace bvg {

using namespace rows;

namespace {

// Threads per workgroup.  The phases that walk global memory are latency-bound and 16 wavefronts hide 4x what 4 do -- but a giant workgroup shares its
// CU with the lean scan kernel, whose wavefronts take 128 registers: 1 024 threads at 99 (104 allocated) registers are 416 of a SIMD's 512, so ONE giant
// workgroup kept every tier-0 wavefront off its CU for its 240 us (most of them spent on the ~15 ordinary nodes behind the large list, at 5 % vector
// activity): the 0.3 % of the default workload's blocks that are giants held 9 % of the chip.  512 threads at 88 registers leave room for two tier-0
// wavefronts per SIMD next to a giant: 251 -> 273 G edges/s on the default workload (256 / 384 / 512 threads: 272.6 / 272.2 / 273.7; 128: 258; 64: 229;
// 1 024 threads squeezed into 64 / 80 registers: 269.7 / 268.4; profiles/r04_ab_giantwg*.txt).
#ifndef BVG_GIANT_THREADS
#define BVG_GIANT_THREADS 512
#endif
#ifndef BVG_GIANT_MINWG
#define BVG_GIANT_MINWG 1                    // (experiments: minimum wavefronts per SIMD, i.e. a register budget)
#endif
constexpr unsigned GNT = BVG_GIANT_THREADS;
constexpr uint32_t kGStageWords = 2048;      // LDS window over the stream: 8 KiB
constexpr uint32_t kHdrMin = 48, kHdrEvery = 16;   // copy-block / interval sections this long get index entries, one per kHdrEvery codes (whatever the residuals' granularity)
typedef MaskPrefix<uint64_t> MP;

__device__ __forceinline__ uint32_t gword_be(const uint8_t* g, uint64_t w) { return __builtin_bswap32(reinterpret_cast<const uint32_t*>(g)[w]); }
// MSB-first windows straight from the .graph bytes (readable 16 bytes past the last record)
__device__ __forceinline__ uint32_t gwin32(const uint8_t* g, uint64_t bit) {
    const uint64_t w = bit >> 5; return funnel(gword_be(g, w), gword_be(g, w + 1), (uint32_t)bit & 31u);
}
__device__ __forceinline__ uint64_t gwin64(const uint8_t* g, uint64_t bit) {
    const uint64_t w = bit >> 5; const uint32_t sh = (uint32_t)bit & 31u;
    const uint32_t a = gword_be(g, w), b = gword_be(g, w + 1), c = gword_be(g, w + 2);
    return ((uint64_t)funnel(a, b, sh) << 32) | funnel(b, c, sh);
}
// a value every lane holds identically, moved to scalar registers: what is computed from it runs on the scalar unit
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | uni32((uint32_t)v); }
// leading zeros of a 64-bit value held in scalar registers (opaque to the optimiser, which would otherwise turn `lz < 32` into a
// 64-bit comparison: a VECTOR instruction whose result the scalar loop has to wait for); 0xFFFFFFFF for 0
__device__ __forceinline__ uint32_t sclz64(uint64_t w) { uint32_t r; asm volatile("" : "+s"(w)); asm("s_flbit_i32_b64 %0, %1" : "=s"(r) : "s"(w)); return r; }   // (pinned to a register pair first: a folded constant is no operand)
// first index in [from, n) at which a monotone predicate turns true (n if never): doubling steps from `from`, then bisection
template <typename F> __device__ __forceinline__ uint32_t gallop_first(uint32_t from, uint32_t n, F pred) {
    uint32_t b = from, st = 1;
    while (b + st <= n && !pred(b + st - 1u)) { b += st; st <<= 1; }
    uint32_t lo = b, hi = b + st - 1u < n ? b + st - 1u : n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (!pred(mid)) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ uint64_t gwave_incl_scan64(uint64_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t t = __shfl_up(v, o, 64); if ((int)lane >= o) v += t; }
    return v;
}
// number of elements <= v in a sorted array
__device__ __forceinline__ uint32_t upper_bound64(const uint64_t* arr, uint32_t n, uint64_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (arr[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}

#ifdef BVG_PROF
#define GP_T(i) do { gp_t[i] = clock64(); } while (0)
#define GP_WHY(n) do { if (tid == 0 && (a.dbg & 1024u)) printf("[giant] block %u node %lld refused at site %d (hd6 %x)\n", bid, (long long)x, n, hd[6]); } while (0)
#define GP_REPORT() do { if (tid == 0 && d > 100000u && (a.dbg & 512u)) printf("[giant] node %lld d %u ref %u bc %u ic %u nres %u | Kcycles: header %lld room %lld residuals %lld Z1a %lld Z1b %lld Z2 %lld\n", (long long)x, d, ref, bc, ic, nres, \
    (long long)(gp_t[1] - gp_t[0]) >> 10, (long long)(gp_t[2] - gp_t[1]) >> 10, (long long)(gp_t[3] - gp_t[2]) >> 10, (long long)(gp_t[4] - gp_t[3]) >> 10, (long long)(gp_t[5] - gp_t[4]) >> 10, (long long)(gp_t[6] - gp_t[5]) >> 10); } while (0)
#else
#define GP_T(i) do {} while (0)
#define GP_WHY(n) do {} while (0)
#define GP_REPORT() do {} while (0)
#endif

template <typename T, bool MAT>
__global__ void __launch_bounds__(GNT, BVG_GIANT_MINWG) giant_kernel(DecodeArgs a) {
#ifdef BVG_PROF
    long long gp_t[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    __shared__ __attribute__((aligned(16))) uint32_t stage[kGStageWords];
    __shared__ uint64_t nd_base[kRing];
    __shared__ uint32_t nd_d[kRing];
    __shared__ uint32_t wg_bad;
    __shared__ uint32_t hd[12];               // wavefront 0's header parse, handed to the others

    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const uint32_t bid = a.work_list ? a.work_list[blockIdx.x] : (a.blk_lo + blockIdx.x);
    const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
    if (e <= a.from || s >= a.to || s >= e) return;
    const uint32_t halo = a.blk_halo[bid];
    const uint64_t hmask = a.blk_mask[bid];
    const uint32_t W = (uint32_t)a.window;
    const int64_t hs = s - (int64_t)halo;
    const int64_t rep_lo = s > a.from ? s : a.from, rep_hi = e < a.to ? e : a.to;

    // the work area: slot blockIdx.x of a batched launch, or -- one launch for all the giants -- a slot taken from the launch's shared set: there are more
    // slots than workgroups can be resident at once, so a free one always turns up, and every holder finishes without waiting for anybody
    __shared__ uint32_t slot_sh;
    if (tid == 0) {
        uint32_t sl = blockIdx.x;
        if (a.gslots) {
            sl %= a.gnslots;
            while (atomicCAS(&a.gslots[sl], 0u, 1u) != 0u) { sl = sl + 1u < a.gnslots ? sl + 1u : 0u; __builtin_amdgcn_s_sleep(2); }
            __threadfence();              // acquire: what the previous holder of the area wrote is visible before this workgroup touches it.  (INVARIANT today: a workgroup never reads
        }                                 //  bytes of pool / scr that it has not written itself, so nothing depends on this yet; a change that inherits contents would.)
        slot_sh = sl;
    }
    __syncthreads();
    const uint32_t slot = slot_sh;
    T* const pool = reinterpret_cast<T*>(a.gpool) + (uint64_t)slot * a.gpool_elems;
    uint64_t* const scr = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(a.gscr) + (uint64_t)slot * a.gscr_elems * sizeof(T));
    const uint64_t CAP = a.gpool_elems < 0xFFFFFFF0ull ? a.gpool_elems : 0xFFFFFFF0ull;    // (positions inside the area are 32-bit)
    const uint64_t SCR = a.gscr_elems * sizeof(T) / sizeof(uint64_t);
    const uint32_t zk = (uint32_t)a.cod.zeta_k, minint = (uint32_t)a.min_interval;
    const bool zfast = zk >= 2;
    const uint32_t kSkipMin = a.skip_min, kSkipShift = a.skip_shift, kSkipEvery = 1u << kSkipShift;   // (this index's granularity: they hide the compile-time defaults of bvg_kernels.h)

    for (unsigned i = tid; i < (unsigned)kRing; i += GNT) { nd_base[i] = 0; nd_d[i] = 0; }
    if (tid == 0) wg_bad = 0;
    __syncthreads();

    // residual skip index: a giant's entry takes two 16-bit slots (its records are longer than 64 Kbit), format 2
    const bool sk_any = a.skip_first != nullptr && !a.batch;
    const uint64_t sk_base = sk_any ? a.skip_first[bid] : 0ull;
    const uint32_t sk_slots = sk_any ? (uint32_t)(a.skip_first[bid + 1] - sk_base) : 0u;
    const bool sk_use = a.skip_mode == 0 && sk_slots != 0 && a.skip_fmt && a.skip_fmt[bid] == 2;
    const bool sk_fill = a.skip_mode == 2 && sk_any;
    uint32_t sk_run = 0;                                       // slots of the nodes walked so far
    if (sk_fill && a.skip_val) {                               // the value slots this format leaves unused read as zero, whatever was there (the dense walk of bvg_index.hip visits every block)
        for (uint32_t i = tid; i < sk_slots; i += GNT) reinterpret_cast<T*>(a.skip_val)[sk_base + i] = (T)0;
        __syncthreads();
    }
    auto rd32 = [&](uint64_t sl) -> uint32_t { return (uint32_t)a.skip_bit[sl] | ((uint32_t)a.skip_bit[sl + 1] << 16); };
    auto wr32 = [&](uint64_t sl, uint32_t val) { a.skip_bit[sl] = (uint16_t)(val & 0xFFFFu); a.skip_bit[sl + 1] = (uint16_t)(val >> 16); };

    uint64_t pool_used = 0;
    uint64_t stg_bit0 = 0; uint32_t stg_bits = 0;
    uint64_t chk = 0, blk_arcs = 0, blk_nodes = 0;             // chk: per thread; arcs / nodes: uniform
    unsigned err = 0;
    bool failed = false;
    uint32_t fail_need = 0xFFFFFFF5u;
    uint64_t cur = 0;                                          // bit cursor of the walk in step (uniform)

    // The walk in step is WAVEFRONT 0's alone (the scalar unit is shared by the four SIMDs of a CU: four wavefronts walking in step
    // would take turns on it); the others wait at the next barrier.  It reads from a 128-bit register buffer (hi: the next 64 bits of the stream, always valid), refilled 64 bits at a
    // time from the LDS window: the LDS round trip is off the per-code dependency chain (a code costs ~100 cycles instead of ~600).
    uint64_t hi = 0, lo = 0; uint32_t avail = 0, widx = 0, stg_words = 0;
    auto seek = [&](uint64_t pos) {                            // (re)position at bit `pos`; restages the window when pos + 192 bits are not in it
        if (!(stg_bits && pos >= stg_bit0 && (pos + 192 <= stg_bit0 + stg_bits || stg_bit0 + stg_bits >= a.padded_bytes * 8ull))) {
            wave_sync();
            const uint64_t b0 = (pos >> 3) & ~15ull;
            uint64_t nb = a.padded_bytes > b0 ? a.padded_bytes - b0 : 0;
            if (nb > kGStageWords * 4ull) nb = kGStageWords * 4ull;
            for (uint32_t c = lane; c < (uint32_t)(nb >> 4); c += 64u) {
                const uint4 v = *reinterpret_cast<const uint4*>(a.graph + b0 + ((uint64_t)c << 4));
                uint4 w; w.x = __builtin_bswap32(v.x); w.y = __builtin_bswap32(v.y); w.z = __builtin_bswap32(v.z); w.w = __builtin_bswap32(v.w);
                *reinterpret_cast<uint4*>(&stage[c << 2]) = w;
            }
            stg_bit0 = uni64(b0 << 3); stg_bits = uni32((uint32_t)(nb << 3)); stg_words = uni32((uint32_t)(nb >> 2));
            wave_sync();
        }
        const uint32_t rel = (uint32_t)(pos - stg_bit0), wi = rel >> 5, sh = rel & 31u;
        const uint32_t m = kGStageWords - 1;                   // (indices past the staged words read stale LDS: never past a record's end)
        hi = uni64(((uint64_t)stage[wi & m] << 32) | stage[(wi + 1) & m]); lo = uni64(((uint64_t)stage[(wi + 2) & m] << 32) | stage[(wi + 3) & m]);
        if (sh) { hi = (hi << sh) | (lo >> (64u - sh)); lo <<= sh; }
        avail = 128u - sh; widx = wi + 4; cur = pos;
    };
    auto consume = [&](uint32_t n) {                            // 1 <= n <= 64
        if (n < 64u) { hi = (hi << n) | (lo >> (64u - n)); lo <<= n; } else { hi = lo; lo = 0; }
        avail -= n; cur += n;
        if (avail < 64u) {
            if (widx + 2u > stg_words) { seek(cur); return; }
            const uint64_t w = uni64(((uint64_t)stage[widx] << 32) | stage[widx + 1]); widx += 2;
            if (avail) { hi |= w >> avail; lo = w << (64u - avail); } else { hi = w; lo = 0; }
            avail += 64u;
        }
    };
    auto rd_gamma = [&](uint64_t& v) -> bool {                  // gamma of a value < 2^31, all on the scalar unit
        const uint32_t lz = sclz64(hi);
        if (lz > 31u) return false;
        const uint32_t l = 2u * lz + 1u;
        v = (hi >> (64u - l)) - 1u;
        consume(l);
        return true;
    };
    // What the walk in step decodes is collected 64 entries at a time in registers (entry i in lane i mod 64) and stored by one wavefront in one coalesced instruction: a store per entry would put a vector comparison and an
    // exec-mask change on every iteration of the scalar loop.
    const uint32_t wv = uni32(tid >> 6);
    uint32_t st_a = 0, st_b = 0, st_c = 0;
    auto put = [&](uint32_t i, uint32_t a32, uint32_t b32, uint32_t c32) {
        const bool mine = lane == (i & 63u);                  // a comparison and a select per value: vector work beside the scalar chain, nothing feeds back
        st_a = mine ? a32 : st_a; st_b = mine ? b32 : st_b; st_c = mine ? c32 : st_c;
    };
    // true for the lanes that store now: entry i was the last of its group of 64 (or the last of all)
    auto group_full = [&](uint32_t i, uint32_t n) -> bool { return (i & 63u) == 63u || i + 1u == n; };   // scalar: the 64 entries are stored now
    // ---- code boundaries resolved in the wavefront (long sections without index entries: first scan, index build).  The next 128 bits
    // of the stream sit in (hi, lo); lane b decodes the code that WOULD start at bit b of them (length and value); the codes that
    // really start there are the chain 0 -> len[0] -> ... followed on the scalar side with v_readlane (five scalar instructions per
    // code instead of the ~45 of the walk in step); the chain's lanes then hold one chunk of up to 64 consecutive codes, and prefix
    // scans over them turn gaps into values.
    auto load128 = [&](uint64_t pos) {
        if (!(stg_bits && pos >= stg_bit0 && (pos + 192 <= stg_bit0 + stg_bits || stg_bit0 + stg_bits >= a.padded_bytes * 8ull))) { seek(pos); }
        const uint32_t rel = (uint32_t)(pos - stg_bit0), wi = rel >> 5, sh = rel & 31u, m = kGStageWords - 1;
        const uint32_t d0 = uni32(stage[wi & m]), d1 = uni32(stage[(wi + 1) & m]), d2 = uni32(stage[(wi + 2) & m]), d3 = uni32(stage[(wi + 3) & m]), d4 = uni32(stage[(wi + 4) & m]);
        hi = ((uint64_t)d0 << 32) | d1; lo = ((uint64_t)d2 << 32) | d3;
        if (sh) { hi = (hi << sh) | (lo >> (64u - sh)); lo = (lo << sh) | ((uint64_t)d4 >> (32u - sh)); }
        cur = pos;
    };
    auto lane_window = [&]() -> uint64_t { return lane ? (hi << lane) | (lo >> (64u - lane)) : hi; };
    // the chain through the per-lane lengths: mask of the lanes where a code starts, their number (at most `want`), the bits they span
    auto follow_chain = [&](uint32_t lenv, uint32_t want, uint64_t& mask, uint32_t& n, uint32_t& span) -> bool {
        mask = 0; n = 0; uint32_t pos = 0;
        while (pos < 64u && n < want) {
            const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lenv, (int)pos);
            if (l == 0) return false;
            mask |= 1ull << pos; pos += l; n++;
        }
        span = pos;
        return true;
    };

    for (int64_t x = hs; x < e; x++) {
        const uint32_t hbit = x < s ? (uint32_t)(s - 1 - x) : 0;
        if (!(x >= s || ((hmask >> hbit) & 1ull))) continue;
        const uint64_t off_x = uni64(a.offsets[x]), rec_end = uni64(a.offsets[x + 1]);
        uint64_t v;
        bool bad = false;
        GP_T(0);
        // ---------------------------------------------------------------- header (BVG:1003-1058) in stages: the counts by wavefront 0 in
        // step; the copy blocks and the intervals in parallel tasks of kHdrEvery codes when the index holds their entries (every
        // kHdrEvery-th block / interval of a long section: bit offset + the running sums), else by wavefront 0 in step as well
        uint32_t d = 0, ref = 0, bc = 0, ic = 0, nres = 0, rlen = 0, ivtot = 0;
        uint64_t rlb = 0;
        // H1: outdegree, reference, block count
        if (wv == 0) {
            uint32_t hfail = 0;
            do {
                seek(off_x);
                if (!rd_gamma(v) || v > 0x7FFFFFFFull) { hfail = 0xFFFFFFF5u; break; }
                d = (uint32_t)v;
                if (d == 0) break;
                if (W > 0) {                                                   // readReference (unary), BVG:692-703
                    const uint32_t lz = sclz64(hi);
                    if (lz >= 63u) { hfail = 0xFFFFFFF5u; break; }
                    consume(lz + 1); v = lz;
                    if (v > W || (int64_t)v > x) { err |= ERR_REF_RANGE; v = 0; }
                    ref = (uint32_t)v;
                }
                if (ref > 0) {
                    if (!rd_gamma(v) || v > rec_end - (cur < rec_end ? cur : rec_end) + 1) { hfail = 0xFFFFFFF5u; break; }
                    bc = (uint32_t)v;
                    if ((uint64_t)bc + 4 > SCR) { hfail = 0xFFFFFFF2u; break; }
                }
            } while (0);
            if (lane == 0) { hd[0] = d; hd[1] = ref; hd[2] = bc; hd[6] = hfail; hd[7] = (uint32_t)cur; hd[8] = (uint32_t)(cur >> 32); }
        }
        __syncthreads();
        d = uni32(hd[0]); ref = uni32(hd[1]); bc = uni32(hd[2]);
        if (uni32(hd[6])) { failed = true; GP_WHY(1); fail_need = uni32(hd[6]); break; }
        if (wv != 0) cur = ((uint64_t)uni32(hd[8]) << 32) | uni32(hd[7]);
        if (ref > 0) { rlen = uni32(nd_d[(uint32_t)(x - ref) & RM]); rlb = uni64(nd_base[(uint32_t)(x - ref) & RM]); }
        int64_t extra = d;
        // H2: copy blocks (BVG:1023-1032) in prefix form
        const uint32_t Eb = bc >= kHdrMin ? (bc - 1u) / kHdrEvery : 0u;       // index entries of this section: 6 slots each
        const uint32_t eb_first = sk_run; sk_run += 6u * Eb;
        if (sk_use && sk_run > sk_slots) { failed = true; GP_WHY(2); break; }             // index out of step with the stream
        if (bc) {
            __syncthreads();                                                   // (hd is rewritten)
            if (sk_use && Eb) {
                for (uint32_t q = tid; q <= Eb; q += GNT) {
                    const uint32_t i0 = q * kHdrEvery, cnt = q == Eb ? bc - i0 : kHdrEvery;
                    uint64_t pos = cur, tot = 0, cop = 0;
                    if (q) { const uint64_t sl = sk_base + eb_first + 6ull * (q - 1u); pos = off_x + rd32(sl); tot = rd32(sl + 2); cop = rd32(sl + 4); if (!(pos > cur && pos < rec_end)) { bad = true; pos = cur; } }
                    for (uint32_t i = 0; i < cnt && !bad; i++) {
                        const uint64_t w = gwin64(a.graph, pos);
                        const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
                        if (lz > 31u) { bad = true; break; }
                        const uint32_t l = 2u * lz + 1u; pos += l;
                        const uint64_t b = (w >> (64u - l)) - 1u + ((i0 + i) ? 1u : 0u);
                        tot += b; if (!((i0 + i) & 1u)) cop += b;
                        scr[i0 + i] = MP::pack((uint32_t)tot, (uint32_t)cop);
                    }
                    if (q == Eb) { hd[0] = (uint32_t)pos; hd[1] = (uint32_t)(pos >> 32); hd[2] = (uint32_t)tot; hd[3] = (uint32_t)(tot >> 32); hd[4] = (uint32_t)cop; hd[5] = (uint32_t)(cop >> 32); }
                }
                if (bad) atomicOr(&wg_bad, 1u);
            } else if (wv == 0) {
                uint64_t tot = 0, copied = 0;
                if (bc >= kHdrMin) {                                           // boundaries resolved in the wavefront, a chunk of codes per step
                    bool lbad = false, cbad = false;                           // (per lane / wave-uniform: the loop's own exits stay uniform)
                    for (uint32_t i = 0; i < bc;) {
                        load128(cur);
                        const uint64_t w = lane_window();
                        const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
                        const uint32_t len = lz < 32u ? 2u * lz + 1u : 0u;
                        const uint32_t val = len ? (uint32_t)((w >> (64u - len)) - 1u) : 0u;
                        uint64_t mask; uint32_t n, span;
                        if (!follow_chain(len, bc - i, mask, n, span)) { cbad = true; break; }
                        const bool on = (mask >> lane) & 1ull;
                        const uint32_t gi = i + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                        const uint32_t bv = on ? val + (gi ? 1u : 0u) : 0u, ev = (on && !(gi & 1u)) ? bv : 0u;
                        const uint32_t ti = wave_incl_scan32(bv), ci = wave_incl_scan32(ev);
                        if (on) {
                            const uint64_t t = tot + ti, c = copied + ci;
                            scr[gi] = MP::pack((uint32_t)t, (uint32_t)c);
                            if (sk_fill && Eb && gi && (gi & (kHdrEvery - 1u)) == 0) {
                                const uint64_t sl = sk_base + eb_first + 6ull * (gi / kHdrEvery - 1u), rel = cur + lane - off_x;
                                if (sl + 5 < sk_base + sk_slots) { wr32(sl, (uint32_t)rel); wr32(sl + 2, (uint32_t)(t - bv)); wr32(sl + 4, (uint32_t)(c - ev)); }
                                if (rel > 0xFFFFFFFFull) lbad = true;
                            }
                        }
                        tot += lane_get(ti, 63); copied += lane_get(ci, 63);
                        i += n; cur += span;
                    }
                    if (cbad || ballot(lbad)) bad = true;
                    seek(cur);                                                 // the walk in step goes on from here
                } else
                // (the loop is bounded by the count, which the record's length bounds: running off the record's end is checked behind it,
                // so that the loop-carried chain stays on the scalar unit -- 64-bit comparisons are vector operations)
                for (uint32_t i = 0; i < bc; i++) {
                    if (sk_fill && Eb && i && (i & (kHdrEvery - 1u)) == 0 && lane == 0) {
                        const uint64_t sl = sk_base + eb_first + 6ull * (i / kHdrEvery - 1u);
                        if (sl + 5 < sk_base + sk_slots) { wr32(sl, (uint32_t)(cur - off_x)); wr32(sl + 2, (uint32_t)tot); wr32(sl + 4, (uint32_t)copied); }
                        if (cur - off_x > 0xFFFFFFFFull) bad = true;
                    }
                    if (!rd_gamma(v)) { bad = true; break; }
                    const uint64_t b = v + (i ? 1u : 0u);
                    tot += b; if (!(i & 1u)) copied += b;
                    put(i, (uint32_t)tot, (uint32_t)copied, 0u);
                    if (group_full(i, bc) && lane <= (i & 63u)) scr[(i & ~63u) + lane] = MP::pack(st_a, st_b);
                }
                if (bad) atomicOr(&wg_bad, 1u);
                if (lane == 0) { hd[0] = (uint32_t)cur; hd[1] = (uint32_t)(cur >> 32); hd[2] = (uint32_t)tot; hd[3] = (uint32_t)(tot >> 32); hd[4] = (uint32_t)copied; hd[5] = (uint32_t)(copied >> 32); }
            }
            __syncthreads();
            if (uni32(wg_bad)) { failed = true; GP_WHY(3); break; }
            cur = ((uint64_t)uni32(hd[1]) << 32) | uni32(hd[0]);
            const uint64_t tot = ((uint64_t)uni32(hd[3]) << 32) | uni32(hd[2]); uint64_t copied = ((uint64_t)uni32(hd[5]) << 32) | uni32(hd[4]);
            if (cur > rec_end || tot > rlen) { failed = true; GP_WHY(4); break; }         // blocks running past the referenced list: the literal kernel decides
            if (!(bc & 1u)) copied += rlen - tot;                              // BVG:1030
            extra = (int64_t)d - (int64_t)copied;
            if (extra < 0) { failed = true; GP_WHY(5); break; }
        } else if (ref > 0) {                                                  // no blocks: the whole referenced list is copied (BVG:1030)
            extra = (int64_t)d - (int64_t)rlen;
            if (extra < 0) { failed = true; GP_WHY(14); break; }
        }
        // H3: interval count (always gamma, BVG:1040)
        const uint64_t ib = bc;                                                // intervals behind the blocks: left[ic], before[ic + 1], position[ic]
        if (d > 0 && extra > 0 && minint != 0) {
            __syncthreads();
            if (wv == 0) {
                uint32_t hfail = 0;
                seek(cur);
                if (!rd_gamma(v) || v > (rec_end - (cur < rec_end ? cur : rec_end)) / 2 + 1) hfail = 0xFFFFFFF5u;
                else if (ib + 3ull * v + 4 > SCR) hfail = 0xFFFFFFF2u;
                if (lane == 0) { hd[3] = hfail ? 0u : (uint32_t)v; hd[6] = hfail; hd[7] = (uint32_t)cur; hd[8] = (uint32_t)(cur >> 32); }
            }
            __syncthreads();
            ic = uni32(hd[3]);
            if (uni32(hd[6])) { failed = true; GP_WHY(6); fail_need = uni32(hd[6]); break; }
            if (wv != 0) cur = ((uint64_t)uni32(hd[8]) << 32) | uni32(hd[7]);
        }
        // H4: intervals (BVG:1042-1058)
        const uint32_t Ei = ic >= kHdrMin ? (ic - 1u) / kHdrEvery : 0u;       // 8 slots each
        const uint32_t ei_first = sk_run; sk_run += 8u * Ei;
        if (sk_use && sk_run > sk_slots) { failed = true; GP_WHY(7); break; }
        if (ic) {
            __syncthreads();
            if (sk_use && Ei) {
                for (uint32_t q = tid; q <= Ei; q += GNT) {
                    const uint32_t i0 = q * kHdrEvery, cnt = q == Ei ? ic - i0 : kHdrEvery;
                    uint64_t pos = cur, before = 0; int64_t prev = 0;
                    if (q) {
                        const uint64_t sl = sk_base + ei_first + 8ull * (q - 1u);
                        pos = off_x + rd32(sl); prev = (int64_t)(((uint64_t)rd32(sl + 4) << 32) | rd32(sl + 2)); before = rd32(sl + 6);
                        if (!(pos > cur && pos < rec_end)) { bad = true; pos = cur; }
                    }
                    for (uint32_t i = 0; i < cnt && !bad; i++) {
                        uint64_t w = gwin64(a.graph, pos);
                        uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
                        if (lz > 31u) { bad = true; break; }
                        uint32_t l = 2u * lz + 1u; pos += l;
                        const uint64_t v1 = (w >> (64u - l)) - 1u;
                        w = gwin64(a.graph, pos); lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
                        if (lz > 31u) { bad = true; break; }
                        l = 2u * lz + 1u; pos += l;
                        const uint64_t v2 = (w >> (64u - l)) - 1u;
                        const int64_t left = (i0 + i) == 0 ? x + nat2int64(v1) : prev + 1 + (int64_t)v1;
                        if ((i0 + i) == 0 && left < 0) bad = true;
                        const int64_t len = (int64_t)v2 + minint;
                        prev = left + len;
                        scr[ib + i0 + i] = (uint64_t)(T)left; scr[ib + ic + i0 + i] = before;
                        before += (uint64_t)len;
                    }
                    if (q == Ei) { scr[ib + 2ull * ic] = before; hd[0] = (uint32_t)pos; hd[1] = (uint32_t)(pos >> 32); hd[2] = (uint32_t)before; hd[3] = (uint32_t)(before >> 32); }
                }
                if (bad) atomicOr(&wg_bad, 1u);
            } else if (wv == 0) {
                int64_t prev = 0, left0 = 0; uint64_t before = 0;
                if (ic >= kHdrMin) {
                    // the 2 * ic gamma codes alternate (gap to the left end, length - minInterval); over their chain, a running sum S of
                    // {1 + gap | length} makes S the left end at a gap code and the interval's end at a length code (BVG:1042-1058)
                    bool lbad = false, cbad = false;                           // (per lane / wave-uniform: `bad` itself may have been set per lane before)
                    const uint32_t ctot = 2u * ic;
                    for (uint32_t cdone = 0; cdone < ctot;) {
                        load128(cur);
                        const uint64_t w = lane_window();
                        const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
                        const uint32_t len = lz < 32u ? 2u * lz + 1u : 0u;
                        const uint64_t val = len ? (w >> (64u - len)) - 1u : 0u;
                        uint64_t mask; uint32_t n, span;
                        if (!follow_chain(len, ctot - cdone, mask, n, span)) { cbad = true; break; }
                        const bool on = (mask >> lane) & 1ull;
                        const uint32_t gc = cdone + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                        const bool isgap = on && !(gc & 1u), islen = on && (gc & 1u);
                        const uint32_t iv = gc >> 1;                           // the interval this code belongs to
                        const int64_t c = !on ? 0 : isgap ? (gc == 0 ? x + nat2int64(val) : 1 + (int64_t)val) : (int64_t)val + minint;
                        const uint64_t lc = islen ? val + minint : 0ull;
                        const int64_t S = prev + (int64_t)gwave_incl_scan64((uint64_t)c, lane);
                        const uint64_t Bf = before + gwave_incl_scan64(lc, lane);   // at a gap code: the elements of the intervals before it
                        if (isgap) {
                            if (gc == 0 && S < 0) lbad = true;
                            scr[ib + iv] = (uint64_t)(T)S; scr[ib + ic + iv] = Bf;
                            if (sk_fill && Ei && iv && (iv & (kHdrEvery - 1u)) == 0) {
                                const uint64_t sl = sk_base + ei_first + 8ull * (iv / kHdrEvery - 1u), rel = cur + lane - off_x; const uint64_t pv = (uint64_t)(S - c);
                                if (sl + 7 < sk_base + sk_slots) { wr32(sl, (uint32_t)rel); wr32(sl + 2, (uint32_t)pv); wr32(sl + 4, (uint32_t)(pv >> 32)); wr32(sl + 6, (uint32_t)Bf); }
                                if (rel > 0xFFFFFFFFull) lbad = true;
                            }
                        }
                        prev = (int64_t)lane_get64((uint64_t)S, 63); before = lane_get64(Bf, 63);
                        cdone += n; cur += span;
                    }
                    if (cbad || ballot(lbad)) bad = true;
                    seek(cur);
                } else
                for (uint32_t i = 0; i < ic; i++) {
                    if (sk_fill && Ei && i && (i & (kHdrEvery - 1u)) == 0 && lane == 0) {
                        const uint64_t sl = sk_base + ei_first + 8ull * (i / kHdrEvery - 1u);
                        if (sl + 7 < sk_base + sk_slots) { wr32(sl, (uint32_t)(cur - off_x)); wr32(sl + 2, (uint32_t)(uint64_t)prev); wr32(sl + 4, (uint32_t)((uint64_t)prev >> 32)); wr32(sl + 6, (uint32_t)before); }
                        if (cur - off_x > 0xFFFFFFFFull) bad = true;
                    }
                    uint64_t v1, v2;
                    if (!rd_gamma(v1) || !rd_gamma(v2)) { bad = true; break; }
                    const int64_t left = i == 0 ? x + nat2int64(v1) : prev + 1 + (int64_t)v1;
                    if (i == 0) left0 = left;
                    const int64_t len = (int64_t)v2 + minint;
                    prev = left + len;
                    put(i, (uint32_t)left, (uint32_t)((uint64_t)left >> 32), (uint32_t)before);
                    if (group_full(i, ic) && lane <= (i & 63u)) { const uint32_t j = (i & ~63u) + lane; scr[ib + j] = (uint64_t)(T)(((uint64_t)st_b << 32) | st_a); scr[ib + ic + j] = st_c; }
                    before += (uint64_t)len;
                }
                if (bad || left0 < 0) atomicOr(&wg_bad, 1u);
                if (lane == 0) { scr[ib + 2ull * ic] = before; hd[0] = (uint32_t)cur; hd[1] = (uint32_t)(cur >> 32); hd[2] = (uint32_t)before; hd[3] = (uint32_t)(before >> 32); }
            }
            __syncthreads();
            if (uni32(wg_bad)) { failed = true; GP_WHY(8); break; }
            cur = ((uint64_t)uni32(hd[1]) << 32) | uni32(hd[0]);
            const uint64_t before = ((uint64_t)uni32(hd[3]) << 32) | uni32(hd[2]);
            extra -= (int64_t)before;
            if (cur > rec_end || before > 0x7FFFFFFFull || extra < 0) { failed = true; GP_WHY(9); break; }
            ivtot = (uint32_t)before;
        }
        if (d > 0) nres = (uint32_t)extra;
        // the residuals follow in step unless index entries cut them into tasks: wavefront 0's buffer goes back to the cursor.  (Round 6, found by fuzz seed 601: the test was
        // `nres >= kSkipMin`, but a list of EXACTLY kSkipMin residuals has no entry -- (nres - 1) >> shift == 0 -- and is decoded in step too: without the seek it read from wherever
        // the header walk had left the buffer, and the self-check at the record's end refused the block on every scan after the index was built.)
        const bool res_tasks = sk_use && nres >= kSkipMin && ((nres - 1u) >> kSkipShift) != 0u;
        if (wv == 0 && nres && !res_tasks) seek(cur);
        const uint32_t cntE = nres >= kSkipMin ? (nres - 1u) >> kSkipShift : 0u;    // index entries of the residuals: 2 slots (+ a value) each
        const uint32_t efirst = sk_run;
        sk_run += 2u * cntE;
        if (a.skip_mode == 1) {                                                 // the index build only counts entries here: the header walk was all it needs
            __syncthreads();
            if (tid == 0) { nd_base[(uint32_t)x & RM] = 0; nd_d[(uint32_t)x & RM] = d; }
            __syncthreads();
            continue;
        }
        const uint64_t* const B = scr; const uint64_t* const IL = scr + bc; const uint64_t* const IC = scr + bc + ic; uint64_t* const IP = scr + bc + 2ull * ic + 1;

        GP_T(1);
        // ---------------------------------------------------------------- room: the list at the bottom, the residuals parked at the top
        if (pool_used + (uint64_t)d + nres + 1 > CAP && pool_used > 0) {
            // keep only the lists of the last W nodes, moved to the front (oldest first: a move never lands on a list not yet moved)
            uint64_t packed = 0;
            for (uint32_t j = W; j >= 1; j--) {
                const int64_t y = x - (int64_t)j;
                if (y < hs) continue;
                const uint64_t src = nd_base[(uint32_t)y & RM]; const uint64_t len = nd_d[(uint32_t)y & RM];
                if (src != packed)
                    for (uint64_t t0 = 0; t0 < len; t0 += GNT) {
                        const uint64_t t = t0 + tid; T vv = 0;
                        if (t < len) vv = pool[src + t];
                        __syncthreads();
                        if (t < len) pool[packed + t] = vv;
                        __syncthreads();
                    }
                __syncthreads();
                if (tid == 0) nd_base[(uint32_t)y & RM] = packed;
                packed += len;
            }
            pool_used = packed;
            __syncthreads();
            if (ref > 0) rlb = nd_base[(uint32_t)(x - ref) & RM];
        }
        if (pool_used + (uint64_t)d + nres + 1 > CAP) { failed = true; GP_WHY(10); fail_need = 0xFFFFFFF2u; break; }
        const uint64_t base = pool_used;
        T* const out = pool + base;
        T* const rt = pool + (CAP - nres - 1);
        const T* const rl = pool + rlb;

        GP_T(2);
        // ---------------------------------------------------------------- residuals (BVG:902-935)
        if (nres > 0) {
            if (sk_use && cntE) {
                if (sk_run > sk_slots) { failed = true; GP_WHY(11); break; }               // index out of step with the stream
                __syncthreads();
                const uint32_t Ttot = cntE + 1u;
                for (uint32_t p0 = 0; p0 < Ttot; p0 += GNT) {
                    const uint32_t q = p0 + tid;
                    if (q < Ttot) {
                        const uint32_t t0 = q << kSkipShift;
                        uint32_t cnt = q == cntE ? nres - t0 : kSkipEvery;
                        uint64_t pos = cur; T r = (T)x;
                        if (q) {
                            const uint64_t sl = sk_base + efirst + 2ull * (q - 1u);
                            const uint32_t rel = (uint32_t)a.skip_bit[sl] | ((uint32_t)a.skip_bit[sl + 1] << 16);
                            pos = off_x + rel;
                            r = sizeof(T) == 8 ? (T)(*reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(a.skip_val) + sl * 8ull))
                                               : (T)(reinterpret_cast<const uint32_t*>(a.skip_val)[sl]);
                            if (!(pos > cur && pos < rec_end)) { bad = true; cnt = 0; }
                        }
                        for (uint32_t i = 0; i < cnt; i++) {
                            uint32_t len = 0; uint64_t val = 0;
                            if (zfast) { uint32_t v32; len = zeta_fast32(gwin32(a.graph, pos), zk, v32); val = v32; }
                            if (len == 0) len = zeta64(gwin64(a.graph, pos), zk, val);
                            if (len == 0) { bad = true; break; }
                            pos += len;
                            r = (t0 + i) == 0 ? (T)(r + (T)nat2int64(val)) : (T)(r + 1 + (T)val);
                            rt[t0 + i] = r;
                            if (pos > rec_end) { err |= ERR_OVERRUN; break; }
                        }
                        if (q == cntE && pos != rec_end && !bad) err |= ERR_MALFORMED;   // SURVEY A.6 self-check
                    }
                }
            } else if (wv == 0) {
                // wavefront 0 in step from the sliding window (64 values stored at a time); the index build records every kSkipEvery-th start
                T r = (T)x;
                if (nres >= kHdrMin) {                                         // boundaries resolved in the wavefront (see load128)
                    bool lbad = false, cbad = false;
                    for (uint32_t t = 0; t < nres;) {
                        load128(cur);
                        const uint64_t w = lane_window();
                        uint32_t len = 0; uint64_t val = 0;
                        if (zfast) { uint32_t v32; len = zeta_fast32((uint32_t)(w >> 32), zk, v32); val = v32; }
                        if (len == 0) len = zeta64(w, zk, val);
                        uint64_t mask; uint32_t n, span;
                        if (!follow_chain(len, nres - t, mask, n, span)) { cbad = true; break; }
                        const bool on = (mask >> lane) & 1ull;
                        const uint32_t gt = t + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                        const uint64_t c = !on ? 0ull : gt == 0 ? (uint64_t)nat2int64(val) : 1ull + val;
                        const uint64_t S = gwave_incl_scan64(c, lane);
                        if (on) {
                            const T rv = (T)(r + (T)S);
                            rt[gt] = rv;
                            if (sk_fill && cntE && gt && (gt & (kSkipEvery - 1u)) == 0) {
                                const uint64_t sl = sk_base + efirst + 2ull * ((gt >> kSkipShift) - 1u), rel = cur + lane - off_x;
                                if (sl + 1 < sk_base + sk_slots) {
                                    wr32(sl, (uint32_t)rel);
                                    const T before_v = (T)(rv - (T)c);
                                    if (sizeof(T) == 8) *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(a.skip_val) + sl * 8ull) = (uint64_t)before_v;
                                    else reinterpret_cast<uint32_t*>(a.skip_val)[sl] = (uint32_t)before_v;
                                }
                                if (rel > 0xFFFFFFFFull) lbad = true;
                            }
                        }
                        r = (T)(r + (T)lane_get64(S, 63));
                        t += n; cur += span;
                    }
                    if (cbad || ballot(lbad)) bad = true;
                } else
                for (uint32_t t = 0; t < nres; t++) {
                    if (sk_fill && cntE && t && (t & (kSkipEvery - 1u)) == 0 && lane == 0) {
                        const uint64_t sl = sk_base + efirst + 2ull * ((t >> kSkipShift) - 1u);
                        if (sl + 1 < sk_base + sk_slots) {
                            const uint64_t rel = cur - off_x;
                            a.skip_bit[sl] = (uint16_t)(rel & 0xFFFFu); a.skip_bit[sl + 1] = (uint16_t)((rel >> 16) & 0xFFFFu);
                            if (sizeof(T) == 8) *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(a.skip_val) + sl * 8ull) = (uint64_t)r;
                            else reinterpret_cast<uint32_t*>(a.skip_val)[sl] = (uint32_t)r;
                            if (rel > 0xFFFFFFFFull) bad = true;
                        }
                    }
                    uint32_t len = 0; uint64_t val = 0;
                    if (zfast) { uint32_t v32; len = zeta_fast32((uint32_t)(hi >> 32), zk, v32); val = v32; }
                    if (len == 0) len = zeta64(hi, zk, val);
                    if (len == 0) { bad = true; break; }
                    consume(len);
                    r = t == 0 ? (T)(r + (T)nat2int64(val)) : (T)(r + 1 + (T)val);
                    put(t, (uint32_t)r, (uint32_t)((uint64_t)r >> 32), 0u);
                    if (group_full(t, nres) && lane <= (t & 63u)) rt[(t & ~63u) + lane] = (T)(((uint64_t)st_b << 32) | st_a);
                }
                if (cur > rec_end) err |= ERR_OVERRUN;
                else if (cur != rec_end && !bad) err |= ERR_MALFORMED;         // SURVEY A.6 self-check
            }
        } else if (wv == 0 && cur != rec_end) err |= ERR_MALFORMED;
        if (tid == 0 && nres + 1u > 0) rt[nres] = sentinel<T>();               // guard behind the residual positions
        if (bad) atomicOr(&wg_bad, 1u);
        __syncthreads();
        if (uni32(wg_bad)) { failed = true; GP_WHY(12); break; }

        GP_T(3);
        // ---------------------------------------------------------------- emission by output position (BVG:1062-1090)
        const bool rep = x >= rep_lo && x < rep_hi;
        uint32_t k0 = 0, k1 = 0;
        if (rep && !MAT) {
            node_key((uint64_t)x + a.node_base, k0, k1);
        }
        bool zbad = false;
        // copied elements below v: rank of its lower bound in the referenced list under the mask; an element equal to one of
        // [v, v + len) would be emitted once by the reference's merge
        auto copied_below = [&](T vv, uint32_t len) -> uint32_t {
            if (!rlen) return 0u;
            const uint32_t qq = lds_lower_bound<T>(rl, rlen, vv);
            uint32_t qn;
            const uint32_t t = MP::rank(B, bc, rlen, qq, qn);
            if (qn < rlen && (T)(rl[qn] - vv) < (T)len) zbad = true;
            return t;
        };
        // Z1a: one thread per interval (the residual VALUES are still in place)
        for (uint32_t q = tid; q < ic; q += GNT) {
            const T vv = (T)IL[q]; const uint32_t len = (uint32_t)(IC[q + 1] - IC[q]);
            const uint32_t lb = lds_lower_bound<T>(rt, nres, vv);
            if (lb < nres && (T)(rt[lb] - vv) < (T)len) zbad = true;           // a residual inside the interval
            const uint64_t pe = IC[q] + lb + copied_below(vv, len);
            if (pe + len > d) { zbad = true; IP[q] = 0; } else IP[q] = pe;
        }
        __syncthreads();
        GP_T(4);
        // Z1b: the residuals in runs of consecutive ones: their places rise with them, so the three searches
        // (intervals at or below the value, its lower bound in the referenced list, the copy block of that position) gallop on from
        // where the residual before ended instead of starting over -- a few probes next to each other instead of ~20 far apart
        {
            // a wavefront takes a run of consecutive residuals, its lanes every 64th of them: coalesced loads and stores, no lane reads a
            // line it has just written, and a lane's next value is only 64 residuals on
            constexpr uint32_t NWV = GNT / 64u;
            const uint32_t R = ((nres + NWV - 1u) / NWV + 63u) & ~63u;
            const uint64_t b0 = (uint64_t)wv * R;
            const uint32_t i0 = b0 < nres ? (uint32_t)b0 : nres, i1 = (uint64_t)i0 + R < nres ? i0 + R : nres;
            uint32_t j = 0, qq = 0, bh = 0;                                     // intervals with left <= v; lower bound in rl; first block ending behind it
            for (uint32_t i = i0 + lane; i < i1; i += 64u) {
                const T vv = rt[i];
                uint64_t pe = i;
                if (ic) {
                    j = gallop_first(j, ic, [&](uint32_t m) { return IL[m] > (uint64_t)vv; });
                    if (j) { const uint64_t cj = IC[j]; pe += cj; if ((uint64_t)vv - IL[j - 1] < cj - IC[j - 1]) zbad = true; }
                }
                if (rlen) {
                    qq = gallop_first(qq, rlen, [&](uint32_t m) { return !(rl[m] < vv); });
                    bh = gallop_first(bh, bc, [&](uint32_t m) { return MP::pos(B[m]) > qq; });
                    const uint64_t prev = bh ? B[bh - 1] : 0ull;
                    const bool keepb = bh < bc ? !(bh & 1u) : !(bc & 1u);       // behind the last block: kept iff their number is even
                    uint32_t qn;
                    if (keepb) { qn = qq; pe += MP::kept(prev) + (qq - MP::pos(prev)); }
                    else { qn = bh < bc ? MP::pos(B[bh]) : rlen; pe += MP::kept(prev); }
                    if (qn < rlen && rl[qn] == vv) zbad = true;                 // a copied element equal to the residual
                }
                if (pe >= d) { zbad = true; pe = 0; }
                out[pe] = vv;
                if (!MAT && rep) chk += mix_node<T>(k1, vv);
                rt[i] = (T)pe;
            }
        }
        __syncthreads();
        GP_T(5);
        // Z2: GNT equal tasks of consecutive output positions
        {
            uint32_t S = (d + GNT - 1u) / GNT; if (S < kMinTask) S = kMinTask;
            uint32_t p = tid * S, pstop = p + S < d ? p + S : d;
            if (tid * (uint64_t)S >= d) { p = 0; pstop = 0; }
            uint32_t ri = 0, rnext = kInf, ivk = ic, ivpos = kInf, ivlen = 0, qcur = 0, krem = kInf, bi = bc;
            T ivleft = 0;
            if (p < pstop) {
                ri = lds_lower_bound<T>(rt, nres, (T)p);                       // residual positions below p
                rnext = ri < nres ? (uint32_t)rt[ri] : kInf;
                uint32_t ie = ivtot;
                if (ic) {                                                      // the first interval ending behind p
                    uint32_t lo = 0, hi = ic;
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (IP[mid] + (IC[mid + 1] - IC[mid]) > p) hi = mid; else lo = mid + 1; }
                    if (lo < ic) { ivk = lo; ivpos = (uint32_t)IP[lo]; ivlen = (uint32_t)(IC[lo + 1] - IC[lo]); ivleft = (T)IL[lo]; ie = (uint32_t)IC[lo] + (p > ivpos ? p - ivpos : 0u); }
                }
                const uint32_t t = p - ri - ie;                                // rank of the next copied element among the kept ones
                if (rlen) MP::select(B, bc, rlen, t, qcur, krem, bi);          // MaskedLongIterator.java:73-100
            }
            const uint32_t rlast = rlen ? rlen - 1u : 0u;
            for (; p < pstop; p++) {
                if (p == rnext) { ri++; rnext = ri < nres ? (uint32_t)rt[ri] : kInf; continue; }   // a residual: placed by Z1b
                const uint32_t io = p - ivpos;
                const bool ii = io < ivlen;                                    // LongIntervalSequenceIterator.java:71-78
                T ov;
                if (ii) {
                    ov = (T)(ivleft + (T)io);
                    if (io + 1u == ivlen) {
                        ivk++; ivpos = kInf; ivlen = 0;
                        if (ivk < ic) { ivpos = (uint32_t)IP[ivk]; ivlen = (uint32_t)(IC[ivk + 1] - IC[ivk]); ivleft = (T)IL[ivk]; }
                    }
                } else {
                    if (!rlen || qcur > rlast) { zbad = true; ov = 0; }
                    else ov = rl[qcur];
                    qcur++;
                    if (--krem == 0) MP::next_block(B, bc, rlen, qcur, krem, bi);   // MaskedLongIterator.java:81-100
                }
                out[p] = ov;
                if (!MAT && rep) chk += mix_node<T>(k1, ov);
            }
        }
        if (zbad) atomicOr(&wg_bad, 1u);
        __syncthreads();
        GP_T(6);
        GP_REPORT();
        if (uni32(wg_bad)) { failed = true; GP_WHY(13); break; }
        if (tid == 0) { nd_base[(uint32_t)x & RM] = base; nd_d[(uint32_t)x & RM] = d; }
        pool_used = base + d;
        if (rep) { blk_arcs += d; blk_nodes += 1; if (!MAT && tid == 0) chk += mix_node_const(k0, k1, a.node_base, d); }
        if (MAT && rep) {                                                       // coalesced copy-out
            const uint64_t dst0 = a.batch ? a.cum[bid >> 1] : a.cum[x - a.from];
            for (uint64_t t = tid; t < d; t += GNT) {
                const T vv = out[t];
                a.succ[dst0 + t] = (int64_t)((uint64_t)vv + a.node_base);
            }
            if (tid == 0 && a.outdeg && !a.batch) a.outdeg[x - a.from] = (int32_t)d;
        }
        __syncthreads();
    }

    __syncthreads();                                            // every thread is done with the work area: hand the slot back
    if (a.gslots && tid == 0) { __threadfence(); atomicExch(&a.gslots[slot], 0u); }
    if (failed) {
        if (tid == 0) {
            uint32_t fslot = atomicAdd(a.fail_count, 1u);
            if (fslot < a.fail_cap) { a.fail_list[fslot] = bid; if (a.fail_need) a.fail_need[fslot] = fail_need; }
        }
        return;
    }
    if (a.skip_mode == 1 && a.skip_cnt && tid == 0) a.skip_cnt[bid] = sk_run;
    if (a.skip_mode == 2 && a.skip_fmt && tid == 0) a.skip_fmt[bid] = 2;
    chk = wave_sum64(chk); err = wave_or32(err);
    if (lane == 0) {
        unsigned long long* const accs = a.acc + (size_t)(bid & a.acc_mask) * kAccStride;   // this block's result stripe
        if (chk) atomicAdd(&accs[1], (unsigned long long)chk);
        if (err) atomicOr(&accs[3], (unsigned long long)err);
        if (tid == 0) {
            if (blk_arcs) atomicAdd(&accs[0], (unsigned long long)blk_arcs);
            if (blk_nodes) atomicAdd(&accs[2], (unsigned long long)blk_nodes);
        }
    }
}

}  // namespace

// scan / materialise the blocks of the work list with one GNT-thread workgroup each; a.gpool / a.gscr: per-workgroup areas as for
// decode_kernel<SLOW>.  Default codings and windows <= kMaxWindow only (the caller checks).
static const size_t giant_pad_default = 0;
void launch_giant_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, bool materialise, hipStream_t s) {
    if (nblocks == 0) return;
    dim3 grid(nblocks), block(GNT);
    // unused LDS per workgroup (experiments): with > 70 KB only ONE giant workgroup fits a CU, which leaves wave slots and registers for the
    // wavefronts of tier 0 launched beside it
    const size_t pad = knob("BVG_GIANT_PAD") ? (size_t)std::min(140000, std::max(0, atoi(knob("BVG_GIANT_PAD")))) : giant_pad_default;
    if (wide) {
        if (materialise) hipLaunchKernelGGL((giant_kernel<uint64_t, true>), grid, block, pad, s, a);
        else hipLaunchKernelGGL((giant_kernel<uint64_t, false>), grid, block, pad, s, a);
    } else {
        if (materialise) hipLaunchKernelGGL((giant_kernel<uint32_t, true>), grid, block, pad, s, a);
        else hipLaunchKernelGGL((giant_kernel<uint32_t, false>), grid, block, pad, s, a);
    }
}

}  // namespace bvg
