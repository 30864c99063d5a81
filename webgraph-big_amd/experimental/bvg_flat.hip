// experimental/bvg_flat.hip — the flat scan kernel (round 5).  EXPERIMENTAL: bit-exact (GPU suite with BVG_FLAT=1, tests/test_emu.py), but SLOWER than
// scan_kernel (eu15 190 vs 276 G edges/s, tiled cnr-2000 98 vs 141 G: profiles/r05_ab_flat2_*.txt): built only by `make experimental`, selected by BVG_FLAT=1.
// Tier 0 and the LDS classes of a steady-state successor scan on 32-bit lists.
//
// Same contract as scan_kernel (bvg_scan.hip): one wavefront per VALIDATED node block (bvg_api.hip build_skip: every consistency check has
// passed once, the stream in HBM is immutable), BVGraph's default codings, scan mode, residual skip index present; a block that does not
// fit fails over to the checking kernels through the same fail list.  What is different is the control structure:
//   * PER-RECORD STATE LIVES IN AN LDS TABLE, not in the registers of "the lane that owns the record".  A super-row is up to R = 64 * G
//     consecutive records (G = 1 on dense graphs; 2-4 on sparse ones, where 64 records hold only a few hundred arcs): their headers are
//     parsed in rounds of 64, one record per lane (outdegree, reference, copy blocks straight into prefix form, intervals: BVG:1003-1060),
//     and every field a later pass needs is written to 16-bit arrays indexed by the record (d, list base, copy blocks, intervals, residual
//     cursor, skip entries ...).  A task of any later pass finds its record by a binary search over a prefix array in LDS and READS the
//     fields -- no pass is tied to 64 records, no parameter travels by ds_bpermute (scan_kernel: 10-14 shuffles per dealt task);
//   * every pass after the parse is ONE FLAT TASK LIST over all records of the sub-row (the records whose stored lists fit the pool
//     together), dealt 64 tasks at a time: residual segments (<= 16 gaps each, cut at the skip entries; long segments first, short tails
//     after them), the extras of the stored lists of a stage (Z1: output position of every residual / interval by rank, as in scan_kernel),
//     and RUN ITEMS;
//   * NO POSITION LOOP.  scan_kernel builds a stored list that copies from another one by stepping through its output positions (one
//     position per lane and step: ~40 instructions, block ends / residual positions / interval ends tracked per step).  Here the list is
//     cut into RUNS once: Z1 leaves, for every extra in value order, {copied elements below it, extra elements up to and including it};
//     a run starts at every kept copy block (MaskedLongIterator.java:73-100) and behind every extra, and ends at the next of either --
//     (source position, destination position, length), found with two binary searches per item.  Runs are then copied like a leaf's: cut
//     into chunks of 8 elements dealt to all lanes, straight-line groups of {4 LDS reads, 4 writes, 4 multiply-adds}.  Leaves (lists nobody
//     copies from: never materialised) put their kept blocks and intervals (LongIntervalSequenceIterator.java:57-78) into the SAME item
//     lists, without a destination: one item pass and one chunk loop per stage serve stored lists and leaves alike;
//   * stages instead of per-sub-row levels: a list without reference is complete after the residual pass (decoded in place, around its
//     intervals); a record that copies from a list of stage s is in stage s + 1, leaf or not.
// Semantics kept from the reference: MaskedLongIterator.java:73-100 (blocks alternate keep / skip, an even number of blocks keeps the
// tail), MergedLongIterator.java:54-92 (the three streams of a record are disjoint in a validated block, so merging is placing),
// BVG:1037-1060 (intervals), BVG:902-935 (residual gaps).  Checksum: include/bvgraph_hip.h (k1(x) * y + k0(x)).
#include "bvg_rows_common.h"

namespace bvg {

namespace {

using namespace rows;

typedef uint32_t T;
constexpr uint32_t kNoList = 0xFFFFu;
constexpr uint32_t kChunk = 8;                 // elements of a run per lane and pass
enum : uint32_t { F_NEED = 1u, F_REP = 2u, F_MARK = 4u, F_DIRECT = 16u, F_D2 = 32u };   // F_MARK: some later record copies from this list: it is STORED
constexpr uint32_t kMaxStage = 30;

// zeta_3 from a 32-bit window (as in bvg_scan.hip): codes of up to 31 bits; returns the length, 0 = take the 64-bit decoder
__device__ __forceinline__ uint32_t zeta3_fast32(uint32_t w, uint32_t& val) {
    const uint32_t h = w ? (uint32_t)__builtin_clz(w) : 32u;
    const uint32_t fits = h <= 6u ? 1u : 0u, hh = fits ? h : 0u;
    const uint32_t h3 = hh + (hh << 1), nb = h3 + 2u, zt = (hh << 2) + 3u;
    const uint32_t t = (w << (hh + 1u)) >> (32u - nb);
    const uint32_t left = 1u << h3;
    const bool shortc = t < left;
    const uint32_t ext = ((t << 1) | ((w >> (31u - zt)) & 1u)) - 1u;
    val = shortc ? t + left - 1u : ext;
    return fits ? zt + (shortc ? 0u : 1u) : 0u;
}
// gamma from the LDS window: codes of up to 31 bits from one 32-bit window, the 64-bit decoder where a lane needs it; 0 = does not fit
__device__ __forceinline__ uint32_t gamma_at(const uint32_t* stage, uint32_t rel, uint64_t& v) {
    const uint32_t w = win32<LIN>(stage, rel);
    const uint32_t lz = w ? (uint32_t)__builtin_clz(w) : 32u;
    if (lz < 16) { const uint32_t len = 2 * lz + 1; v = (w >> (32u - len)) - 1u; return len; }
    return gamma64(win64<LIN>(stage, rel), v);
}
// Dealing: p[0 .. n] is a non-decreasing prefix array (p[0] = 0) in LDS; the owner of task t (t < p[n]) is the last i with p[i] <= t.  `top` = the
// largest power of two <= n (wave-uniform).  Every lane runs the same number of steps; lanes without a task pass t = 0.
__device__ __forceinline__ uint32_t owner_of(const uint16_t* p, uint32_t n, uint32_t top, uint32_t t) {
    uint32_t lo = 0;
    for (uint32_t st = top; st; st >>= 1) {
        const uint32_t m = lo + st;
        const uint32_t v = p[m <= n ? m : n];
        lo = (m <= n && v <= t) ? m : lo;
    }
    return lo < n ? lo : n - 1u;
}
__device__ __forceinline__ uint32_t pow2_floor(uint32_t n) { return n ? 1u << (31u - (uint32_t)__builtin_clz(n)) : 0u; }

// The t-th kept element of a copy mask in prefix form (MaskPrefix, bvg_rows_common.h): its position q in the referenced list, how many kept
// elements its block still holds from there (0 = there is no t-th kept element), and whether it is the FIRST element of its block.
__device__ __forceinline__ void kept_at(const T* blk, uint32_t bc, uint32_t rlen, uint32_t t, uint32_t& q, uint32_t& krem, bool& first) {
    uint32_t lo = 0, hi = bc;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (MaskPrefix<T>::kept(blk[mid]) > t) hi = mid; else lo = mid + 1; }
    const T prev = lo ? blk[lo - 1] : (T)0;
    const uint32_t kp = MaskPrefix<T>::kept(prev), pp = MaskPrefix<T>::pos(prev);
    first = t == kp;
    q = pp + (t - kp);
    if (lo < bc) krem = MaskPrefix<T>::kept(blk[lo]) - t;                 // inside keep block `lo`
    else if (!(bc & 1u)) krem = q < rlen ? rlen - q : 0u;                  // the tail behind an even number of blocks is kept
    else krem = 0u;
}

struct Tab {                                    // the record table: 16-bit / 8-bit arrays over [-wc, R), indexed by the record's place in the super-row
    uint16_t *d, *base, *rtb, *sb, *bc, *ib, *ic, *nres, *ef, *rel, *rec;
    uint8_t *ref, *fl, *st;
    uint16_t *pre, *pre2;
};

template <bool Z3, int OCC>
__global__ void __launch_bounds__(64, OCC) flat_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];     // table | area (lists, parked residuals, scratch) | stream window
    const unsigned lane = threadIdx.x;
    const uint32_t wi = xcd_order(blockIdx.x, gridDim.x, a.xcds);
    const uint32_t bid = a.work_list ? a.work_list[wi] : (a.blk_lo + wi);
    const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
    if (e <= a.from || s >= a.to || s >= e) return;
    const uint32_t halo = a.blk_halo[bid];
    const uint64_t hmask = a.blk_mask[bid];
    const uint32_t W = (uint32_t)a.window;
    const int64_t hs = s - (int64_t)halo;
    const int64_t rep_lo = s > a.from ? s : a.from, rep_hi = e < a.to ? e : a.to;

    const uint32_t R = a.flat_recs, wc = (W + 7u) & ~7u, NT = wc + R;
    Tab tb;
    {
        uint16_t* p16 = reinterpret_cast<uint16_t*>(dyn_lds) + wc;
        tb.d = p16; tb.base = p16 + NT; tb.rtb = p16 + 2 * NT; tb.sb = p16 + 3 * NT; tb.bc = p16 + 4 * NT; tb.ib = p16 + 5 * NT; tb.ic = p16 + 6 * NT;
        tb.nres = p16 + 7 * NT; tb.ef = p16 + 8 * NT; tb.rel = p16 + 9 * NT; tb.rec = p16 + 10 * NT;
        uint8_t* p8 = dyn_lds + 22 * NT + wc;
        tb.ref = p8; tb.fl = p8 + NT; tb.st = p8 + 2 * NT;
        tb.pre = reinterpret_cast<uint16_t*>(dyn_lds + 25 * NT); tb.pre2 = tb.pre + (R + 8);
    }
    const uint32_t tbytes = (25u * NT + 4u * (R + 8u) + 15u) & ~15u;
    T* const pool = reinterpret_cast<T*>(dyn_lds + tbytes);
    T* const scr = pool;                                                        // (one area: lists bottom up, parked residuals below the scratch, scratch at the top)
    const uint32_t CAP = a.lds_pool_elems + a.lds_scr_elems;
    uint32_t* const stage_w = reinterpret_cast<uint32_t*>(pool + CAP);
    const uint32_t* const stage = stage_w;
    const uint32_t stage_bits = a.lds_stage_words * 32u;
    const uint32_t zk = (uint32_t)a.cod.zeta_k, minint = (uint32_t)a.min_interval;
    const bool zfast = zk >= 2;
    const uint32_t kSkipMin = a.skip_min, kSkipShift = a.skip_shift, kSkipEvery = 1u << kSkipShift;
    const uint64_t nbase = a.node_base;

    for (int i = (int)lane - (int)wc; i < 0; i += 64) { tb.d[i] = 0; tb.base[i] = (uint16_t)kNoList; tb.fl[i] = 0; }
    wave_sync();

    uint32_t pool_used = 0;
    uint64_t stg_bit0 = 0; uint32_t stg_bits = 0;
    uint64_t blk_arcs = 0, blk_chk = 0, blk_nodes = 0;
    unsigned err = 0;
    bool failed = false;
    uint32_t fail_need = 0xFFFFFFFFu;
    const uint64_t sk_base = a.skip_first[bid];
    const uint32_t sk_n = (uint32_t)(a.skip_first[bid + 1] - sk_base);
    uint32_t sk_run = 0;
    // -DBVG_FLAT_PROF (`make flatprof`; BVG_DBG=64 prints the slots): wave-cycles per section {0 super-row set-up, 1 header rounds, 2 peek / marks, 3 sub-row sizing + stages,
    // 4 residual set-up, 5 residual steps, 6 Z1, 7 item set-up, 8 chunks, 9 compaction} and work counts {10 super-rows, 11 sub-rows, 12 records, 13 residual passes, 14 residual steps,
    // 15 Z1 passes, 16 item passes, 17 chunk passes, 18 chunk steps (x4 elements)}
#ifdef BVG_FLAT_PROF
    uint32_t cyc[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define FT0() ((uint32_t)clock64())
#define FT1(i, t) do { cyc[i] += (uint32_t)clock64() - (t); } while (0)
#define FC(i, n) do { cyc[i] += (uint32_t)(n); } while (0)
#else
#define FT0() 0u
#define FT1(i, t) do { (void)(t); } while (0)
#define FC(i, n) do { } while (0)
#endif

    // keeps only the stored lists of the W records before record `upto` (an index of the current super-row; negative: carried over), at the bottom of the pool
    auto compact = [&](int upto) {
        uint32_t my_d = 0, my_base = 0; const int y = upto - (int)W + (int)lane;
        const bool livelane = lane < W && y >= -(int)wc;
        if (livelane) { my_base = tb.base[y]; my_d = my_base == kNoList ? 0u : tb.d[y]; }
        const uint32_t nincl = wave_incl_scan32(my_d);
        const uint32_t nb = nincl - my_d;
        for (uint32_t jn = 0; jn < W && jn < 64; jn++) {
            const uint32_t src = lane_get(my_base, jn), dst = lane_get(nb, jn), len = lane_get(my_d, jn);
            if (src != dst && len)
                for (uint32_t t0 = 0; t0 < len; t0 += 64) {                     // (moves down over itself: every element of a step is read before any is written)
                    const uint32_t t = t0 + lane; T vv = 0;
                    if (t < len) vv = pool[src + t];
                    wave_sync();
                    if (t < len) pool[dst + t] = vv;
                    wave_sync();
                }
        }
        if (livelane && my_base != kNoList) tb.base[y] = (uint16_t)nb;
        pool_used = lane_get(nincl, 63);
        wave_sync();
    };

    int64_t r0 = hs;
    while (r0 < e && !failed) {
        // ================================================================== SUPER-ROW: up to R records, parsed in rounds of 64
        const uint32_t left = (uint32_t)(e - r0 > (int64_t)R ? (int64_t)R : e - r0);
        const uint32_t tq0 = FT0(); FC(10, 1);
        if (pool_used > 0) compact(0);
        FT1(9, tq0);
        const uint32_t SCRH = (CAP - pool_used) >> 1;                           // copy blocks and intervals of the super-row: at most half of what the carried lists leave
        uint32_t scr_used = 0, K = 0;
        bool bad = false, peek_ok = false, unknown = false;
        uint32_t pk0 = 0, pk1 = 0, pk2 = 0; uint64_t pk_off = 0; bool pk_have = false; uint32_t pk_first = 64;   // the next super-row's first records, peeked at (below)
        for (uint32_t g = 0; g * 64u < left && !failed; g++) {
            const uint32_t tqr = FT0();
            const uint32_t r = g * 64u + lane;
            const int64_t x = r0 + (int64_t)r;
            const bool valid = r < left;
            uint64_t o0 = 0, o1 = 0;
            if (valid) { o0 = a.offsets[x]; o1 = a.offsets[x + 1]; }
            const uint32_t nv = left - g * 64u < 64u ? left - g * 64u : 64u;
            if (g == 0) {                                                       // (re)stage the window when the first round's records are not covered by it
                const uint64_t row_lo = lane_get64(o0, 0), row_hi = lane_get64(o1, nv - 1);
                if (!(row_lo >= stg_bit0 && row_hi + 96 <= stg_bit0 + stg_bits)) {
                    wave_sync();
                    const uint64_t b0 = (row_lo >> 3) & ~15ull;
                    uint64_t nb = a.padded_bytes > b0 ? a.padded_bytes - b0 : 0;
                    if (nb > (stage_bits >> 3)) nb = stage_bits >> 3;
                    for (uint32_t c = lane; c < (uint32_t)(nb >> 4); c += 64) {
                        const uint4 v = *reinterpret_cast<const uint4*>(a.graph + b0 + ((uint64_t)c << 4));
                        uint4 w; w.x = __builtin_bswap32(v.x); w.y = __builtin_bswap32(v.y); w.z = __builtin_bswap32(v.z); w.w = __builtin_bswap32(v.w);
                        *reinterpret_cast<uint4*>(&stage_w[c << 2]) = w;
                    }
                    stg_bit0 = b0 << 3; stg_bits = (uint32_t)(nb << 3);
                    wave_sync();
                }
            }
            const bool inwin = valid && o1 + 96 <= stg_bit0 + stg_bits && o0 >= stg_bit0;
            uint32_t kg;
            { const uint64_t m = ballot(inwin); kg = m == ~0ull ? 64u : (uint32_t)__ffsll((unsigned long long)~m) - 1u; if (kg > nv) kg = nv; }
            if (kg == 0) { if (g == 0) { failed = true; fail_need = 0xFFFFFFF1u; } break; }    // a single record larger than the window
            // The records behind a round that the window cut short are the next super-row's first ones, and which of THIS super-row's last W lists
            // they copy from decides what is stored: their first 12 bytes are fetched now and looked at after the parse (their offsets are at hand).
            if (kg < nv && W > 0) {
                pk_first = kg; pk_off = o0;
                const uint64_t pb = (o0 >> 5) << 2;
                if (lane >= kg && lane < kg + W && valid && pb + 12 <= a.padded_bytes) {
                    const uint32_t* gp = reinterpret_cast<const uint32_t*>(a.graph + pb);
                    pk0 = gp[0]; pk1 = gp[1]; pk2 = gp[2]; pk_have = true;
                }
            }
            FT1(0, tqr);
            const uint32_t tqh = FT0();
            uint32_t rel = (uint32_t)(o0 - stg_bit0);
            const uint32_t pend = (uint32_t)(o1 - stg_bit0), recrel = rel;
            const uint32_t hbit = x < s ? (uint32_t)(s - 1 - x) : 0;
            const bool needed = lane < kg && (x >= s || ((hmask >> hbit) & 1ull));
            uint64_t v;
            uint32_t d = 0;
            if (needed) {                                                       // readOutdegree, BVG:654-660
                const uint32_t l = gamma_at(stage, rel, v);
                bad |= l == 0 || v > 0xFFFEull; rel += l; d = bad ? 0u : (uint32_t)v;     // (a list of 2^16 - 1 successors or more does not belong in LDS: the block fails over)
            }
            if (lane < kg) tb.d[r] = (uint16_t)d;
            wave_sync();
            uint32_t ref = 0, bc = 0, ic = 0, nres = 0, sb = 0, ib = 0;
            int32_t extra = (int32_t)d;
            bool parse = needed && d > 0;
            if (parse) {
                if (W > 0) {                                                    // readReference, BVG:692-703
                    const uint64_t w = win64<LIN>(stage, rel);
                    const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
                    bad |= lz >= 64; rel += lz + 1;
                    uint32_t rv = lz;
                    if (rv > W || (int64_t)rv > x) { err |= ERR_REF_RANGE; rv = 0; }
                    ref = rv;
                }
                if (ref > 0) {                                                  // readBlockCount, BVG:728-735
                    const uint32_t l = gamma_at(stage, rel, v);
                    bad |= l == 0 || v > pend - rel + 1; rel += l; bc = bad ? 0u : (uint32_t)v;
                }
            }
            {   // the copy blocks go to the scratch area at the top, this round's below the rounds before
                const uint32_t bincl = wave_incl_scan32(bc > SCRH ? SCRH + 1 : bc);
                const uint32_t kb = (uint32_t)__popcll(ballot(scr_used + bincl <= SCRH));
                if (kb < kg) kg = kb;
                if (kg == 0) { if (g == 0) { failed = true; fail_need = 0xFFFFFFF3u; } break; }
                sb = CAP - scr_used - bincl;
                scr_used += lane_get(bincl, kg - 1);
            }
            parse = parse && lane < kg;
            uint32_t rlenN = 0;
            if (parse) {
                if (ref > 0) {                                                  // copy blocks, BVG:1023-1032: two per step, straight into prefix form
                    uint32_t copied = 0, tot = 0;
                    for (uint32_t i = 0; i < bc; i += 2) {
                        const uint64_t w = win64<LIN>(stage, rel);
                        const uint32_t lz1 = w ? (uint32_t)__builtin_clzll(w) : 64u;
                        const bool two = i + 1u < bc;
                        const uint32_t l1 = 2u * (lz1 & 15u) + 1u;
                        const uint64_t w2 = w << l1;
                        const uint32_t lz2 = w2 ? (uint32_t)__builtin_clzll(w2) : 64u;
                        if (lz1 >= 16u || (two && lz2 >= 16u) || rel > pend) { bad = true; bc = i; break; }
                        const uint32_t l2 = 2u * lz2 + 1u;
                        const uint32_t b1 = (uint32_t)(w >> (64u - l1)) - (i ? 0u : 1u);
                        const uint32_t b2 = (uint32_t)(w2 >> (64u - (l2 & 63u)));
                        tot += b1; copied += b1;
                        scr[sb + i] = MaskPrefix<T>::pack(tot, copied);
                        if (two) { tot += b2; scr[sb + i + 1u] = MaskPrefix<T>::pack(tot, copied); }
                        rel += l1 + (two ? l2 : 0u);
                    }
                    rlenN = tb.d[(int)r - (int)ref];
                    if (tot > rlenN) { bad = true; tot = rlenN; }
                    if (!(bc & 1)) copied += rlenN - tot;                       // BVG:1030
                    extra = (int32_t)d - (int32_t)copied;
                    if (extra < 0) bad = true;
                }
                if (extra > 0 && minint != 0) {                                 // interval count: always gamma (BVG:1040)
                    const uint32_t l = gamma_at(stage, rel, v);
                    bad |= l == 0 || v > (pend - rel) / 2 + 1; rel += l; ic = bad ? 0u : (uint32_t)v;
                }
            }
            {
                const uint32_t iw = lane < kg ? 2 * ic : 0u;
                const uint32_t iincl = wave_incl_scan32(iw > SCRH ? SCRH + 1 : iw);
                const uint32_t ki = (uint32_t)__popcll(ballot(scr_used + iincl <= SCRH));
                if (ki < kg) kg = ki;
                if (kg == 0) { if (g == 0) { failed = true; fail_need = 0xFFFFFFF4u; } break; }
                ib = CAP - scr_used - iincl;
                scr_used += lane_get(iincl, kg - 1);
            }
            parse = parse && lane < kg;
            if (parse) {
                if (ic > 0) {                                                   // intervals, BVG:1042-1058
                    uint32_t prev = 0, big = 0;
                    for (uint32_t i = 0; i < ic; i++) {
                        uint64_t v1, v2;
                        const uint32_t l1 = gamma_at(stage, rel, v1);
                        const uint32_t l2 = gamma_at(stage, rel + l1, v2);
                        if (l1 == 0 || l2 == 0 || rel > pend) { bad = true; ic = i; break; }
                        rel += l1 + l2;
                        big |= (uint32_t)(v1 >> 32) | (uint32_t)(v2 >> 15);          // (an interval entry holds 16 bits of length)
                        const uint32_t u1 = (uint32_t)v1;
                        const uint32_t leftv = i == 0 ? (uint32_t)x + ((u1 >> 1) ^ (0u - (u1 & 1u))) : prev + 1u + u1;   // nat2int, modulo 2^32
                        const uint32_t len = (uint32_t)v2 + minint;
                        prev = leftv + len;
                        extra -= (int32_t)len;
                        bad |= extra < 0 || len > 0xFFFFu;
                        scr[ib + 2 * i] = (T)leftv; scr[ib + 2 * i + 1] = (T)len;
                    }
                    if (extra < 0 || big != 0) { bad = true; extra = 0; }
                }
                nres = (uint32_t)extra;
            }
            if (ballot(bad && lane < kg)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
            // the table entries of the round; then the marks: a record that copies says so on the list it copies from
            const bool rep = needed && lane < kg && x >= rep_lo && x < rep_hi;
            const uint32_t cntE = (parse && nres >= kSkipMin) ? (nres - 1u) >> kSkipShift : 0u;
            {
                const uint32_t eincl = wave_incl_scan32(cntE);
                const uint32_t efirst = sk_run + eincl - cntE;
                sk_run += lane_get(eincl, 63);
                if (lane < kg) {
                    tb.base[r] = (uint16_t)kNoList; tb.sb[r] = (uint16_t)sb; tb.bc[r] = (uint16_t)bc; tb.ib[r] = (uint16_t)ib; tb.ic[r] = (uint16_t)ic;
                    tb.nres[r] = (uint16_t)nres; tb.ef[r] = (uint16_t)efirst; tb.rel[r] = (uint16_t)rel; tb.rec[r] = (uint16_t)recrel;
                    tb.ref[r] = (uint8_t)ref; tb.fl[r] = (uint8_t)((needed ? F_NEED : 0u) | (rep ? F_REP : 0u)); tb.st[r] = 0;
                }
            }
            if (sk_run > sk_n || sk_n > 0xFFFFu) { failed = true; fail_need = 0xFFFFFFF5u; break; }   // index out of step with the stream
            wave_sync();
            if (parse && ref > 0) atomicOr(reinterpret_cast<unsigned*>(tb.fl - wc) + (((int)r - (int)ref + (int)wc) >> 2), F_MARK << ((((int)r - (int)ref + (int)wc) & 3) * 8));
            if (rep) {                                                          // what the node adds besides k1 * (its successors): d * (k1 * base + k0)
                uint32_t k0, k1; node_key((uint64_t)x + nbase, k0, k1);
                blk_chk += mix_node_const(k0, k1, nbase, d); blk_arcs += d; blk_nodes += 1;
            }
            K += kg;
            FT1(1, tqh);
            if (kg < nv) { peek_ok = kg == pk_first; break; }                   // (a round cut by the scratch area after the window cut it: the peeked records are not the next ones)
        }
        if (failed) break;
        wave_sync();
        const uint32_t tqp = FT0(); FC(12, K);
        // ---- which lists are STORED: those with a mark.  The last W records may be copied from by the NEXT super-row's first W records (not by the
        //      next block's: that one decodes its halo itself): peek at their outdegree and reference codes (BVG:654-660, 692-703) -- in the bytes
        //      fetched above, or in the window when the super-row ended on a full round; a record that cannot be read marks all W.
        if (W > 0 && r0 + (int64_t)K < e) {
            uint32_t tgt = 0xFFFFFFFFu;
            uint32_t pi = 64;                                                   // my peeked record is record K + pi
            uint64_t win = 0; bool have = false;
            if (peek_ok) {
                if (lane >= pk_first && lane < pk_first + W) {
                    pi = lane - pk_first;
                    if (r0 + (int64_t)K + pi < e) {
                        if (pk_have) {
                            const uint32_t sh = (uint32_t)pk_off & 31u;
                            const uint64_t hi = ((uint64_t)__builtin_bswap32(pk0) << 32) | __builtin_bswap32(pk1);
                            win = sh ? (hi << sh) | (uint64_t)(__builtin_bswap32(pk2) >> (32u - sh)) : hi; have = true;
                        } else unknown = true;
                    } else pi = 64;
                }
                if (pk_first + W > 64 && r0 + (int64_t)K + (64 - pk_first) < e) unknown = true;      // (some of the W lie beyond the lanes that hold offsets)
            } else if (K == left || (K & 63u) != 0) {
                unknown = true;                                                 // cut by the scratch area: assume every one of the last W is copied from
            } else {                                                            // the super-row ended on a full round: the next records mostly lie in the window
                if (lane < W && r0 + (int64_t)K + lane < e) {
                    pi = lane;
                    const uint64_t po = a.offsets[r0 + (int64_t)K + lane];
                    if (po >= stg_bit0 && po + 160 <= stg_bit0 + stg_bits) { win = win64<LIN>(stage, (uint32_t)(po - stg_bit0)); have = true; }
                    else unknown = true;
                }
            }
            if (have) {
                uint64_t pv;
                const uint32_t l = gamma64(win, pv);
                if (l == 0 || l > 40) unknown = true;
                else if (pv != 0) {
                    const uint64_t w2 = win << l;
                    const uint32_t lz = w2 ? (uint32_t)__builtin_clzll(w2) : 64u;
                    if (lz >= 64u - l) unknown = true;                          // (the unary code runs past the bits at hand)
                    else if (lz > pi && lz <= W && lz - pi <= K) tgt = K + pi - lz;
                }
            }
            if (ballot(unknown)) {
                for (uint32_t j = lane; j < W && j < K; j += 64) tb.fl[K - 1 - j] |= (uint8_t)F_MARK;
            } else if (tgt != 0xFFFFFFFFu) atomicOr(reinterpret_cast<unsigned*>(tb.fl - wc) + ((tgt + wc) >> 2), F_MARK << (((tgt + wc) & 3) * 8));
            wave_sync();
        }
        const uint32_t CAPe = CAP - scr_used;                                   // what is left for the lists and the parked residuals of the sub-rows
        // lists without reference that are stored are decoded in place: DIRECT (no intervals: the list IS its residuals) or around their intervals (D2:
        // interval k starts, by default, behind every residual and the intervals before it; a residual that passes it moves it down)
        for (uint32_t j = lane; j < K; j += 64) {
            uint32_t fl = tb.fl[j];
            if ((fl & F_MARK) && tb.ref[j] == 0 && tb.d[j] != 0) {
                const uint32_t ic = tb.ic[j];
                fl |= ic ? F_D2 : F_DIRECT;
                tb.fl[j] = (uint8_t)fl;
                if (ic) { const uint32_t ib = tb.ib[j]; uint32_t pre = tb.nres[j]; for (uint32_t kk = 0; kk < ic; kk++) { const uint32_t ln = (uint32_t)scr[ib + 2 * kk + 1] & 0xFFFFu; scr[ib + 2 * kk + 1] = (T)(ln | (pre << 16)); pre += ln; } }
            }
        }
        wave_sync();

        FT1(2, tqp);
        // ================================================================== SUB-ROWS [sa, se): the records whose stored lists (and parked residuals) fit the pool together
        uint32_t sa = 0;
        while (sa < K) {
            uint32_t se = sa;
            const uint32_t tqs = FT0(); FC(11, 1);
            {
                const uint32_t avail = CAPe - pool_used;
                uint32_t acc_s = 0, acc_r = 0;
                for (uint32_t c = sa; c < K; c += 64) {
                    const uint32_t j = c + lane; const bool in = j < K;
                    uint32_t size = 0, rsz = 0, nr = 0, icj = 0; bool stored = false, withref = false;
                    if (in) {
                        const uint32_t fl = tb.fl[j]; stored = (fl & F_MARK) != 0 && tb.d[j] != 0;
                        if (stored) { size = tb.d[j]; withref = tb.ref[j] != 0; if (withref) { nr = tb.nres[j]; icj = tb.ic[j]; rsz = 2 * nr + icj + 1; } }
                    }
                    const uint32_t sincl = wave_incl_scan32(size), rincl = wave_incl_scan32(rsz);
                    const bool fits = in && acc_s + acc_r + sincl + rincl <= avail;
                    const uint32_t nf = (uint32_t)__popcll(ballot(fits));
                    if (fits) {
                        const uint32_t rtb = CAPe - acc_r - rincl;
                        tb.base[j] = (uint16_t)(stored ? pool_used + acc_s + sincl - size : kNoList);
                        tb.rtb[j] = (uint16_t)rtb;
                        if (withref) pool[rtb + 2 * nr + icj] = 0xFFFFu;        // the sentinel behind the list's cut points (Z1)
                    }
                    if (nf) { acc_s += lane_get(sincl, nf - 1); acc_r += lane_get(rincl, nf - 1); }
                    se += nf;
                    if (nf < 64) break;
                }
                if (se == sa) {                                                 // the first record alone overflows the pool
                    failed = true;
                    uint32_t d0 = tb.d[sa]; const uint32_t n0 = tb.nres[sa];
                    d0 += 2 * (n0 > d0 ? d0 : n0) + tb.ic[sa] + 1u;
                    fail_need = d0 + pool_used + (d0 >> 2) + 64;
                    break;
                }
                pool_used += acc_s;
            }
            const uint32_t n = se - sa;                                         // records of the sub-row
            const uint32_t top = pow2_floor(n);
            wave_sync();
            // ---- stages: 0 = complete after the residual pass (decoded in place); a record that copies from a list of stage s (or from one of an earlier
            //      sub-row: stage 0) is in stage s + 1; lists decoded around their intervals get those filled in by stage 1
            uint32_t smax = 0;
            {
                for (uint32_t c = sa; c < se; c += 64) {
                    const uint32_t j = c + lane;
                    if (j < se) { const uint32_t fl = tb.fl[j]; tb.st[j] = (uint8_t)(((fl & F_NEED) && tb.d[j]) ? ((fl & F_DIRECT) ? 0u : 1u) : 0u); }
                }
                wave_sync();
                for (uint32_t it = 0; it < kMaxStage + 2; it++) {
                    bool ch = false;
                    for (uint32_t c = sa; c < se; c += 64) {
                        const uint32_t j = c + lane;
                        if (j < se) {
                            const uint32_t ref = tb.ref[j];
                            if (ref && j >= sa + ref) { const uint32_t want = (uint32_t)tb.st[j - ref] + 1u; if (want > tb.st[j]) { tb.st[j] = (uint8_t)want; ch = true; } }
                        }
                    }
                    wave_sync();
                    if (!ballot(ch)) break;
                }
                uint32_t sm = 0;
                for (uint32_t c = sa; c < se; c += 64) { const uint32_t j = c + lane; if (j < se && tb.st[j] > sm) sm = tb.st[j]; }
                smax = wave_max32(sm);
                if (smax > kMaxStage) { failed = true; fail_need = 0xFFFFFFF5u; break; }
            }

            FT1(3, tqs);
            // ------------------------------------------------------------------ residuals (ResidualLongIterator, BVG:902-935): one task per segment of <= 2^shift gaps
            {
                uint32_t NL = 0, NS = 0;
                const uint32_t tqa = FT0();
                for (uint32_t c = sa; c < se; c += 64) {
                    const uint32_t j = c + lane; uint32_t Tn = 0, Sn = 0;
                    if (j < se) {
                        const uint32_t nres = tb.nres[j];
                        if (nres) {
                            const uint32_t ce = nres >= kSkipMin ? (nres - 1u) >> kSkipShift : 0u;
                            const bool shortt = nres - (ce << kSkipShift) <= kShortTask;
                            Tn = ce + (shortt ? 0u : 1u); Sn = shortt ? 1u : 0u;
                        }
                    }
                    const uint32_t ti = wave_incl_scan32(Tn), si = wave_incl_scan32(Sn);
                    if (j < se) { tb.pre[j - sa + 1] = (uint16_t)(NL + ti); tb.pre2[j - sa + 1] = (uint16_t)(NS + si); }
                    NL += lane_get(ti, 63); NS += lane_get(si, 63);
                }
                if (lane == 0) { tb.pre[0] = 0; tb.pre2[0] = 0; }
                wave_sync();
                const uint32_t Ttot = NL + NS;
                uint64_t csum = 0; bool tbad = false;
                FT1(4, tqa);
                for (uint32_t p0 = 0; p0 < Ttot; p0 += 64) {
                    const uint32_t tqb = FT0(); FC(13, 1);
                    const uint32_t t = p0 + lane;
                    const bool tl = t < Ttot, isl = t < NL;
                    const bool anylong = p0 < NL, anyshort = p0 + 63u >= NL && NL < Ttot;
                    const uint32_t io = anylong ? owner_of(tb.pre, n, top, (tl && isl) ? t : 0u) : 0u;
                    const uint32_t is = anyshort ? owner_of(tb.pre2, n, top, (tl && !isl) ? t - NL : 0u) : 0u;
                    const uint32_t i = isl ? io : is, j = sa + i;
                    const uint32_t nres = tb.nres[j], fl = tb.fl[j];
                    const uint32_t t_ce = nres >= kSkipMin ? (nres - 1u) >> kSkipShift : 0u;
                    const uint32_t q = tl ? (isl ? t - tb.pre[i] : t_ce) : 0u;
                    const uint32_t t0 = q << kSkipShift;
                    uint32_t cnt = tl ? (q == t_ce ? nres - t0 : kSkipEvery) : 0u;
                    const uint32_t t_rel = tb.rel[j], t_pend = stg_bits;
                    uint32_t trel = t_rel; T r = (T)((uint32_t)r0 + j);
                    const bool tfirst = q == 0;
                    if (tl && q) {
                        const uint64_t ei = sk_base + tb.ef[j] + q - 1u;
                        trel = (uint32_t)tb.rec[j] + a.skip_bit[ei];
                        r = reinterpret_cast<const T*>(a.skip_val)[ei];
                        if (!(trel > t_rel && trel < t_pend)) { tbad = true; cnt = 0; trel = 0; }
                    }
                    const bool stored = (fl & F_MARK) != 0, inplace = (fl & (F_DIRECT | F_D2)) != 0;
                    const uint32_t taddr = (tl && stored) ? (inplace ? (uint32_t)tb.base[j] : (uint32_t)tb.rtb[j]) + t0 : kInf;
                    uint32_t k1 = 0;
                    if (tl && (fl & F_REP)) { uint32_t k0; node_key((uint64_t)(r0 + (int64_t)j) + nbase, k0, k1); }
                    // lists decoded in place around their intervals: the next interval a task has not passed yet, and what the passed ones add to its positions
                    uint32_t ivl = kInf, ivn = 0, ivk = 0, ioff = 0, tic = 0, tib = 0;
                    const bool anyd2 = ballot(tl && (fl & F_D2) && cnt) != 0;
                    if (tl && (fl & F_D2) && cnt) {
                        tic = tb.ic[j]; tib = tb.ib[j];
                        if (q) while (ivk < tic && scr[tib + 2 * ivk] < r) { ioff += (uint32_t)scr[tib + 2 * ivk + 1] & 0xFFFFu; ivk++; }
                        if (ivk < tic) { ivl = (uint32_t)scr[tib + 2 * ivk]; ivn = (uint32_t)scr[tib + 2 * ivk + 1] & 0xFFFFu; }
                    }
                    FT1(4, tqb);
                    const uint32_t tqc = FT0();
                    for (uint32_t st = 0;; st++) {
                        bool on = st < cnt;
                        if (!ballot(on)) break;
                        FC(14, 1);
                        const uint32_t w32 = win32<LIN>(stage, trel);
                        uint32_t v32 = 0, len = Z3 ? zeta3_fast32(w32, v32) : (zfast ? zeta_fast32(w32, zk, v32) : 0u);
                        uint64_t val = v32;
                        if (ballot(on && len == 0)) {                             // codes longer than 31 bits (or zeta_1): a rare, wave-uniform detour
                            if (on && len == 0) { len = zeta64(win64<LIN>(stage, trel), zk, val); if (len == 0) { tbad = true; cnt = 0; on = false; } }
                        }
                        const T gap = (tfirst && st == 0) ? (T)nat2int64(val) : (T)(1 + (T)val);
                        const T rn = (T)(r + gap);
                        if (anyd2 && ballot(on && (uint32_t)rn > ivl)) {          // this residual passes an interval (or several): it starts right here
                            while (on && (uint32_t)rn > ivl) {
                                scr[tib + 2 * ivk + 1] = (T)(ivn | ((t0 + st + ioff) << 16));
                                ioff += ivn; ivk++;
                                if (ivk < tic) { ivl = (uint32_t)scr[tib + 2 * ivk]; ivn = (uint32_t)scr[tib + 2 * ivk + 1] & 0xFFFFu; } else ivl = kInf;
                            }
                        }
                        if (on && taddr != kInf) pool[taddr + st + ioff] = rn;
                        csum += (uint64_t)(on ? k1 : 0u) * (uint64_t)rn;
                        r = rn; trel = on ? trel + len : trel;
                    }
                    tbad |= trel > t_pend;
                    FT1(5, tqc);
                }
                blk_chk += csum;
                if (ballot(tbad)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
                wave_sync();
            }

            // ------------------------------------------------------------------ stages 1 .. smax
            for (uint32_t S = 1; S <= smax; S++) {
                // ---------------- Z1: one task per extra (residual or interval) of the stage's stored lists with a reference: its output position
                //                  = (extras below it) + (copied elements below it) -- the rank of its lower bound in the referenced list under the copy mask
                {
                    const uint32_t tqz = FT0();
                    uint32_t Itot = 0;
                    for (uint32_t c = sa; c < se; c += 64) {
                        const uint32_t j = c + lane; uint32_t In = 0;
                        if (j < se && tb.st[j] == S && (tb.fl[j] & F_MARK) && tb.ref[j]) In = (uint32_t)tb.nres[j] + tb.ic[j];
                        const uint32_t ii = wave_incl_scan32(In);
                        if (j < se) tb.pre[j - sa + 1] = (uint16_t)(Itot + ii);
                        Itot += lane_get(ii, 63);
                    }
                    wave_sync();
                    for (uint32_t p0 = 0; p0 < Itot; p0 += 64) {
                        FC(15, 1);
                        const uint32_t t = p0 + lane; const bool tl = t < Itot;
                        const uint32_t i = owner_of(tb.pre, n, top, tl ? t : 0u), j = sa + i;
                        const uint32_t q = tl ? t - tb.pre[i] : 0u;
                        if (tl) {
                            const uint32_t ref = tb.ref[j], t_ic = tb.ic[j], t_ib = tb.ib[j], t_nres = tb.nres[j], t_bc = tb.bc[j], t_sb = tb.sb[j];
                            const uint32_t t_rlb = tb.base[(int)j - (int)ref], t_rlen = tb.d[(int)j - (int)ref], t_rtb = tb.rtb[j], t_ob = tb.base[j];
                            const T* const rl = pool + t_rlb; const T* const rt = pool + t_rtb; T* const M = pool + t_rtb + t_nres;
                            T vv; uint32_t len = 1, eb, m;
                            if (q < t_ic) {                                       // interval q: the intervals and residuals below it
                                vv = scr[t_ib + 2 * q]; len = (uint32_t)scr[t_ib + 2 * q + 1] & 0xFFFFu;
                                eb = 0;
                                for (uint32_t k = 0; k < q; k++) eb += (uint32_t)scr[t_ib + 2 * k + 1] & 0xFFFFu;
                                const uint32_t rb = lds_lower_bound<T>(rt, t_nres, vv);
                                eb += rb; m = q + rb;
                            } else {                                              // residual q - ic
                                const uint32_t ri = q - t_ic;
                                vv = rt[ri]; eb = ri; m = ri;
                                for (uint32_t k = 0; k < t_ic; k++) {
                                    const T leftv = scr[t_ib + 2 * k];
                                    if (leftv <= vv) { eb += (uint32_t)scr[t_ib + 2 * k + 1] & 0xFFFFu; m++; }
                                }
                            }
                            uint32_t kb = 0;
                            if (t_rlen) { uint32_t qn; kb = MaskPrefix<T>::rank(scr + t_sb, t_bc, t_rlen, lds_lower_bound<T>(rl, t_rlen, vv), qn); }
                            const uint32_t pe = eb + kb;
                            if (q < t_ic) scr[t_ib + 2 * q + 1] = (T)(len | (pe << 16));
                            else pool[t_ob + pe] = vv;
                            M[m] = (T)(kb | ((eb + len) << 16));                  // cut point: {copied elements below the extra, extra elements up to and including it}
                        }
                    }
                    wave_sync();
                    FT1(6, tqz);
                }
                // ---------------- run items of the stage: for a STORED list with a reference one per kept copy block (the run from its first element to the next
                //                  extra or the block's end), one per extra (the run behind it, unless a kept block starts there) and one per interval (an iota);
                //                  for a list decoded in place its intervals; for a LEAF its kept blocks and its intervals, summed only
                {
                    uint32_t tqi = FT0();
                    uint32_t Q = 0;
                    for (uint32_t c = sa; c < se; c += 64) {
                        const uint32_t j = c + lane; uint32_t Ln = 0;
                        if (j < se && tb.st[j] == S) {
                            const uint32_t fl = tb.fl[j], ref = tb.ref[j], ic = tb.ic[j];
                            const uint32_t nk = ref ? ((uint32_t)tb.bc[j] + 2u) >> 1 : 0u;
                            if (fl & F_MARK) Ln = ref ? nk + tb.nres[j] + 2 * ic : ((fl & F_D2) ? ic : 0u);
                            else if (fl & F_REP) Ln = nk + ic;
                        }
                        const uint32_t li = wave_incl_scan32(Ln);
                        if (j < se) tb.pre[j - sa + 1] = (uint16_t)(Q + li);
                        Q += lane_get(li, 63);
                    }
                    wave_sync();
                    uint64_t lsum = 0;
                    for (uint32_t d0 = 0; d0 < Q; d0 += 64) {
                        FC(16, 1);
                        const uint32_t t = d0 + lane; const bool dl = t < Q;
                        const uint32_t i = owner_of(tb.pre, n, top, dl ? t : 0u), j = sa + i;
                        uint32_t e_src = 0, e_len = 0, e_dst = kInf, e_k1 = 0; bool e_iota = false;
                        if (dl) {
                            const uint32_t q = t - tb.pre[i];
                            const uint32_t fl = tb.fl[j], ref = tb.ref[j], t_ic = tb.ic[j], t_ib = tb.ib[j], t_bc = tb.bc[j], t_sb = tb.sb[j], t_nres = tb.nres[j];
                            const bool stored = (fl & F_MARK) != 0;
                            const uint32_t nk = ref ? (t_bc + 2u) >> 1 : 0u, nx = (stored && ref) ? t_nres + t_ic : 0u;
                            const uint32_t t_ob = tb.base[j];
                            if (fl & F_REP) { uint32_t k0; node_key((uint64_t)(r0 + (int64_t)j) + nbase, k0, e_k1); }
                            if (q < nk + nx) {
                                const uint32_t t_rlb = tb.base[(int)j - (int)ref], t_rlen = tb.d[(int)j - (int)ref];
                                const T* const blk = scr + t_sb; const T* const M = pool + tb.rtb[j] + t_nres;
                                if (q < nk) {                                     // kept block 2q (the tail behind an even number of blocks included)
                                    const uint32_t bi = 2u * q;
                                    const T pv = bi ? blk[bi - 1u] : (T)0;
                                    const uint32_t st0 = MaskPrefix<T>::pos(pv), kk = MaskPrefix<T>::kept(pv);
                                    const uint32_t en0 = bi < t_bc ? MaskPrefix<T>::pos(blk[bi]) : t_rlen;
                                    uint32_t L = en0 > st0 ? en0 - st0 : 0u;
                                    e_src = t_rlb + st0;
                                    if (stored) {                                 // the extras at or before its first element; the run ends at the next one
                                        uint32_t lo = 0, hi = nx;
                                        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (((uint32_t)M[mid] & 0xFFFFu) <= kk) lo = mid + 1; else hi = mid; }
                                        const uint32_t E = lo ? (uint32_t)M[lo - 1] >> 16 : 0u, tn = (uint32_t)M[lo] & 0xFFFFu;
                                        if (tn - kk < L) L = tn - kk;
                                        e_dst = t_ob + kk + E;
                                    }
                                    e_len = L;
                                } else {                                          // the run behind extra q - nk
                                    const uint32_t m = q - nk;
                                    const uint32_t cm = (uint32_t)M[m], tt = cm & 0xFFFFu, tn = (uint32_t)M[m + 1] & 0xFFFFu;
                                    if (tn > tt) {
                                        uint32_t qq, krem; bool first;
                                        kept_at(blk, t_bc, t_rlen, tt, qq, krem, first);
                                        if (!first && krem) { e_src = t_rlb + qq; e_len = tn - tt < krem ? tn - tt : krem; e_dst = t_ob + tt + (cm >> 16); }
                                    }
                                }
                            } else {                                              // interval q - nk - nx: an iota (placed by Z1, or by the residuals that passed it)
                                const uint32_t k = q - nk - nx;
                                const T pk = scr[t_ib + 2 * k + 1];
                                e_src = (uint32_t)scr[t_ib + 2 * k]; e_len = (uint32_t)pk & 0xFFFFu; e_iota = true;
                                if (stored) e_dst = t_ob + ((uint32_t)pk >> 16);
                            }
                        }
                        // the items of the pass, cut into chunks of kChunk elements dealt to all lanes
                        const uint32_t nch = (e_len + kChunk - 1u) / kChunk;
                        const uint32_t cincl = wave_incl_scan32(nch), cs = cincl - nch, Ctot = lane_get(cincl, 63);
                        FT1(7, tqi);
                        const uint32_t tqk = FT0();
                        for (uint32_t p0 = 0; p0 < Ctot; p0 += 64) {
                            FC(17, 1);
                            const bool tl = p0 + lane < Ctot;
                            uint32_t own = 0;                                     // the lane whose item chunk p0 + lane belongs to: the first whose prefix exceeds it
                            for (uint32_t step = 32; step; step >>= 1) {
                                const uint32_t vv = (uint32_t)__shfl((int)cincl, (int)(own + step - 1), 64);
                                own += vv <= p0 + lane ? step : 0u;
                            }
                            const int sl = (tl && own < 64) ? (int)own : (int)lane;
                            const uint32_t s_cs = (uint32_t)__shfl((int)cs, sl, 64);              // (every lane takes part in a shuffle: never in the arm of a ?:)
                            const uint32_t q = tl ? p0 + lane - s_cs : 0u;
                            const uint32_t c_src = __shfl(e_src, sl, 64), c_len = __shfl(e_len, sl, 64), c_dst = __shfl(e_dst, sl, 64), c_k1 = __shfl(e_k1, sl, 64);
                            const bool iota = __shfl((int)(e_iota ? 1 : 0), sl, 64) != 0;
                            const uint32_t o = q * kChunk;
                            const uint32_t nn = tl ? (c_len - o < kChunk ? c_len - o : kChunk) : 0u;
                            const uint32_t b0 = c_src + o;
                            const T* const src = pool + (iota ? 0u : b0);
                            const bool wr = tl && c_dst != kInf;
                            T* const dst = pool + (wr ? c_dst + o : 0u);
                            const uint32_t nmax = wave_max32(nn);
                            FC(18, (nmax + 3u) >> 2);
                            for (uint32_t k = 0; k < nmax; k += 4) {
                                const T v0 = src[k], v1 = src[k + 1], v2 = src[k + 2], v3 = src[k + 3];   // (reads past a run stay inside the LDS allocation: the window lies behind the area)
                                const T u0 = iota ? (T)(b0 + k) : v0, u1 = iota ? (T)(b0 + k + 1) : v1, u2 = iota ? (T)(b0 + k + 2) : v2, u3 = iota ? (T)(b0 + k + 3) : v3;
                                if (wr) { if (k < nn) dst[k] = u0; if (k + 1 < nn) dst[k + 1] = u1; if (k + 2 < nn) dst[k + 2] = u2; if (k + 3 < nn) dst[k + 3] = u3; }
                                lsum += (uint64_t)(k < nn ? c_k1 : 0u) * u0;
                                lsum += (uint64_t)(k + 1 < nn ? c_k1 : 0u) * u1;
                                lsum += (uint64_t)(k + 2 < nn ? c_k1 : 0u) * u2;
                                lsum += (uint64_t)(k + 3 < nn ? c_k1 : 0u) * u3;
                            }
                        }
                        FT1(8, tqk);
                        tqi = FT0();
                    }
                    FT1(7, tqi);
                    blk_chk += lsum;
                    wave_sync();
                }
            }
            if (failed) break;
            sa = se;
            if (sa < K) { const uint32_t tqx = FT0(); compact((int)sa); FT1(9, tqx); }
        }
        if (failed) break;
        // ---- next super-row: the last W records move to the front of the table (what a later record needs of a list it copies from: length and place)
        {
            uint32_t cd = 0, cb = kNoList;
            const int y = (int)K - (int)W + (int)lane;
            if (lane < W && y >= -(int)wc) { cd = tb.d[y]; cb = ((tb.fl[y] & F_MARK) && cd) ? tb.base[y] : kNoList; }
            wave_sync();
            if (lane < W) { tb.d[(int)lane - (int)W] = (uint16_t)cd; tb.base[(int)lane - (int)W] = (uint16_t)cb; tb.fl[(int)lane - (int)W] = (uint8_t)(cb != kNoList ? F_MARK : 0u); }
            wave_sync();
        }
        r0 += K;
    }

    err = wave_or32(err);
    if (failed) {
        if (lane == 0) {
            uint32_t slot = atomicAdd(a.fail_count, 1u);
            if (slot < a.fail_cap) { a.fail_list[slot] = bid; if (a.fail_need) a.fail_need[slot] = fail_need; }
        }
        return;
    }
    blk_arcs = wave_sum64(blk_arcs); blk_chk = wave_sum64(blk_chk); blk_nodes = wave_sum64(blk_nodes);
    if (lane == 0) {
        unsigned long long* const accs = a.acc + (size_t)(bid & a.acc_mask) * kAccStride;
        atomicAdd(&accs[0], (unsigned long long)blk_arcs);
        atomicAdd(&accs[1], (unsigned long long)blk_chk);
        atomicAdd(&accs[2], (unsigned long long)blk_nodes);
        if (err) atomicOr(&accs[3], (unsigned long long)err);
#ifdef BVG_FLAT_PROF
        if (a.dbg & 64u) for (int i = 0; i < 19; i++) atomicAdd(&a.acc[9 + i], (unsigned long long)cyc[i]);
#endif
    }
}

}  // namespace

// bytes of LDS the record table takes in front of the area (R records per super-row, window W)
size_t flat_table_bytes(uint32_t recs, int window) {
    const uint32_t wc = ((uint32_t)window + 7u) & ~7u, NT = wc + recs;
    return (25u * (size_t)NT + 4u * (recs + 8u) + 15u) & ~(size_t)15;
}

void launch_flat_decode(const DecodeArgs& a, uint32_t nblocks, bool many_waves, hipStream_t s) {
    if (nblocks == 0) return;
    const size_t dyn = flat_table_bytes(a.flat_recs, a.window) + (size_t)(a.lds_pool_elems + a.lds_scr_elems + a.lds_stage_words) * 4;
    const bool z3 = a.cod.zeta_k == 3;
    if (many_waves) { if (z3) hipLaunchKernelGGL((flat_kernel<true, 6>), dim3(nblocks), dim3(64), dyn, s, a); else hipLaunchKernelGGL((flat_kernel<false, 6>), dim3(nblocks), dim3(64), dyn, s, a); }
    else { if (z3) hipLaunchKernelGGL((flat_kernel<true, 4>), dim3(nblocks), dim3(64), dyn, s, a); else hipLaunchKernelGGL((flat_kernel<false, 4>), dim3(nblocks), dim3(64), dyn, s, a); }
}

}  // namespace bvg
