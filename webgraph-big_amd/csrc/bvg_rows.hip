// bvg_rows.hip — the LDS-resident row kernel (tiers 0 and 1 of the decode).
//
// One wavefront (a 64-thread workgroup) owns one node block and walks it in ROWS of up to 64
// consecutive nodes, one node per lane.  Everything the row touches lives in LDS:
//   * a linear window over the .graph stream (coalesced 16 B/lane loads, byte-swapped once), read
//     with the lean decoders of bvg_lds_codes.h (two/three ds_read_b32 + funnel shift per code);
//   * the list pool: the successor lists of the row and of the <= W nodes before it (compacted when
//     full, so a block may be arbitrarily long at a bounded footprint);
//   * a scratch area with the row's copy blocks and intervals.
// Phase 1 (parse): every lane decodes its own record — outdegree gamma, reference unary, copy blocks
// gamma, intervals gamma, residual gaps zeta_k turned into absolute values at the tail of the
// node's own list (BVGraph.java:1003-1064); long residual lists are cut at their skip-index entries
// into tasks dealt to all lanes.  Phase 2 (emit), chosen per row: a lock-step data-flow loop — per
// iteration each lane emits one successor by the three-way merge {masked copy of the referenced
// list, intervals, residuals} of BVGraph.java:1062-1090; a lane whose referenced list belongs to a
// lower lane of the same row waits on that lane's `produced` counter, so reference chains pipeline —
// or (TASK variant) the level-synchronous emission by output position described at its code below.
// The LDS footprint is kept as small as the graph allows because resident waves per CU — not HBM
// bandwidth — bound this kernel (profiles/README.md).
//
// Whatever does not fit (a list larger than the pool, a record larger than the window, a code
// longer than 64 bits) makes the block fail over to the next tier (bigger LDS, then the generic
// global-memory kernel in bvg_kernels.hip).
#include "bvg_rows_common.h"

#include <cstdlib>
#include <type_traits>

#ifndef BVG_ROWS_WAVES
#define BVG_ROWS_WAVES 5
#endif
#ifndef BVG_TASK_WAVES
#define BVG_TASK_WAVES 5
#endif

namespace bvg {

namespace {

using namespace rows;

template <typename T, bool MAT, bool GEN, bool TASK>
__global__ void __launch_bounds__(64, TASK ? BVG_TASK_WAVES : BVG_ROWS_WAVES) rows_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];     // pool | scratch | stream window
    __shared__ uint32_t nd_base[kRing];
    __shared__ uint32_t nd_d[kRing];
    __shared__ uint32_t rtmap[64 * (kResUnroll > 2 ? kResUnroll : 2)];   // residual segments -> lanes (skip index): kResUnroll tasks per lane and pass
    uint32_t* const produced = rtmap + 64;        // (phase 2 only: the residual tasks of phase 1 are done by then)             // read/written with wavefront-scope relaxed atomics: plain ds_read/ds_write that
                                                  // the compiler may not cache (a `volatile` here compiles to flat sc0 sc1 + vmcnt(0))

    const unsigned lane = threadIdx.x;
    const uint32_t wi = xcd_order(blockIdx.x, gridDim.x, a.xcds);                      // neighbouring blocks on one XCD (bvg_rows_common.h)
    const uint32_t bid = a.work_list ? a.work_list[wi] : (a.blk_lo + wi);
    const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
    if (e <= a.from || s >= a.to || s >= e) return;
    const uint32_t halo = a.blk_halo[bid];
    const uint64_t hmask = a.blk_mask[bid];
    const uint32_t W = (uint32_t)a.window;
    const int64_t hs = s - (int64_t)halo;
    const int64_t rep_lo = s > a.from ? s : a.from, rep_hi = e < a.to ? e : a.to;

    T* const pool = reinterpret_cast<T*>(dyn_lds);
    T* const scr = pool + a.lds_pool_elems;
    // The stream window is only read while a row is PARSED (outdegrees, headers, residual gaps); the row's lists are only written
    // while it is EMITTED.  The task variant therefore lays the window over the part of the pool the row's lists will take
    // (right behind the lists of the previous W nodes; the residuals of the row are parked at the far end): 2-4 KiB of LDS less
    // per wavefront, i.e. one more resident wavefront per CU -- and throughput is linear in those (profiles/r02/ldspad.sh).
    constexpr bool OVL = TASK;
    constexpr uint32_t kAl = 16 / sizeof(T);                                 // elements per 16 bytes (the window is filled with 16-byte stores)
    const uint32_t SWE = OVL ? (uint32_t)((a.lds_stage_words * 4u + sizeof(T) - 1) / sizeof(T)) : 0u;   // window size in pool elements
    const uint32_t* stage = reinterpret_cast<const uint32_t*>(scr + a.lds_scr_elems);
    uint32_t* stage_w = const_cast<uint32_t*>(stage);
    uint32_t stage_off = 0;                                                  // OVL: first pool element of the window
    bool ovl_dirty = false;                                                  // OVL: a list of the last row reached into the window
    const uint32_t CAP = a.lds_pool_elems, SCR = a.lds_scr_elems;
    const uint32_t stage_bits = a.lds_stage_words * 32u;
    const uint32_t zk = (uint32_t)a.cod.zeta_k, minint = (uint32_t)a.min_interval;
    const bool zfast = !GEN && zk >= 2;
    const uint32_t kSkipMin = a.skip_min, kSkipShift = a.skip_shift, kSkipEvery = 1u << kSkipShift;   // (this index's granularity: they hide the compile-time defaults of bvg_kernels.h)
    constexpr bool LEAN = !MAT;                              // scan mode: unreferenced lists are not materialised

    for (unsigned i = lane; i < (unsigned)kRing; i += 64) { nd_base[i] = 0; nd_d[i] = 0; }
    wave_sync();

    uint32_t pool_used = 0;
    uint64_t stg_bit0 = 0; uint32_t stg_bits = 0;             // staged window (wave-uniform)
    uint64_t blk_arcs = 0, blk_chk = 0, blk_nodes = 0;
    unsigned err = 0;
    bool failed = false;
    uint32_t fail_need = 0xFFFFFFFFu;                        // pool elements that would have been enough (when known)
    uint32_t cnt_iter = 0, cnt_pass = 0, cnt_rows = 0, cnt_tasks = 0, cnt_seek = 0, cnt_leaf = 0, cnt_leafp = 0;   // BVG_DBG & 64: work counters (wave-uniform)
    // -DBVG_PROF builds only (`make prof`): wave-cycles per section {row prep, level prep, task set-up, seeks, merge loop,
    // phase 1, row set-up, headers, pool sizing, residuals}, reported with BVG_DBG & 64.  Off by default: the accumulators cost
    // registers.  Wave-cycles measure latency, not issue slots: sections that other resident waves overlap look larger than they cost.
#ifdef BVG_PROF
    uint32_t cyc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    // a block lives well under 2^32 cycles
#define BVG_T0() ((uint32_t)clock64())
#define BVG_T1(i, t) do { cyc[i] += (uint32_t)clock64() - (t); } while (0)
#else
#define BVG_T0() 0u
#define BVG_T1(i, t) do { (void)(t); } while (0)
#endif

    // residual skip index: entries of this block (sk_n > 0: use them; skip_mode 1/2: count / fill)
    const bool sk_have = a.skip_first != nullptr && !a.batch;
    const uint64_t sk_base = sk_have ? a.skip_first[bid] : 0ull;
    const bool sk_mine = a.skip_mode != 0 || !a.skip_fmt || a.skip_fmt[bid] == 1 || a.skip_fmt[bid] == 3;   // (entries filled by the giant kernel have another layout)
    bool validated = true;                                   // skip_mode 2: every row went through the position logic, which refuses what a well-formed stream never holds
    const uint32_t sk_n = sk_have && sk_mine ? (uint32_t)(a.skip_first[bid + 1] - sk_base) : 0u;
    const bool sk_track = a.skip_mode != 0 || sk_n != 0;
    uint32_t sk_run = 0;

    int64_t r0 = hs;
    // offsets of the first row (later rows are prefetched while the previous row is decoded)
    uint64_t off_x = 0, rec_end = 0;
    if (r0 + lane < e) { off_x = a.offsets[r0 + lane]; rec_end = a.offsets[r0 + lane + 1]; }

    while (r0 < e) {
        // ------------------------------------------------------------------ row set-up
        const uint32_t tq5 = BVG_T0();
        const int64_t x = r0 + lane;
        const bool in_range = x < e;
        const uint32_t hbit = x < s ? (uint32_t)(s - 1 - x) : 0;
        const bool needed = in_range && (x >= s || ((hmask >> hbit) & 1ull));
        const uint32_t left = (uint32_t)(e - r0 > 64 ? 64 : e - r0);
        auto compact = [&]() {
            // keep only the lists of the last W nodes, moved to the front of the pool
            uint32_t my_d = 0, my_base = 0; const int64_t y = r0 - (int64_t)W + (int64_t)lane;
            const bool livelane = lane < W && y >= hs;
            if (livelane) { my_d = nd_d[(uint32_t)y & RM]; my_base = nd_base[(uint32_t)y & RM]; }
            const uint32_t nincl = wave_incl_scan32(my_d);
            const uint32_t nbase = nincl - my_d;
            for (uint32_t jn = 0; jn < W && jn < 64; jn++) {
                const uint32_t src = lane_get(my_base, jn), dst = lane_get(nbase, jn), len = lane_get(my_d, jn);
                if (src != dst)
                    for (uint32_t t0 = 0; t0 < len; t0 += 64) {   // (the lists move down over themselves 64 elements at a time: every element of a step is read before any is written)
                        const uint32_t t = t0 + lane; T vv = 0;
                        if (t < len) vv = pool[src + t];
                        wave_sync();
                        if (t < len) pool[dst + t] = vv;
                        wave_sync();
                    }
            }
            if (livelane) nd_base[(uint32_t)y & RM] = nbase;
            pool_used = lane_get(nincl, 63);
            wave_sync();
        };
        if (OVL) {
            if (pool_used > 0) compact();                                     // every row starts from the window lists alone
            const uint32_t noff = (pool_used + kAl - 1) & ~(kAl - 1);
            if (noff + SWE > CAP) { failed = true; fail_need = pool_used + SWE + (pool_used >> 2) + 64; break; }   // the window lists leave no room
            if (noff != stage_off || ovl_dirty) stg_bits = 0;                 // the previous row's lists were written over the window
            stage_off = noff; ovl_dirty = false;
            stage_w = reinterpret_cast<uint32_t*>(pool + stage_off); stage = stage_w;
        }
        {   // (re)stage the window when this row's records are not covered by it
            const uint64_t row_lo = lane_get64(off_x, 0);
            const uint64_t row_hi = lane_get64(rec_end, left - 1);
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
        // rows are cut where the records stop fitting the window (the next row restages from there)
        const bool inwin = in_range && rec_end + 96 <= stg_bit0 + stg_bits && off_x >= stg_bit0;
        uint32_t kwin = (uint32_t)__popcll(ballot(inwin) & (left == 64 ? ~0ull : ((1ull << left) - 1)));
        {   // contiguous prefix only
            const uint64_t m = ballot(inwin);
            kwin = m == ~0ull ? 64u : (uint32_t)__ffsll((unsigned long long)~m) - 1u;
            if (kwin > left) kwin = left;
        }
        if (kwin == 0) { failed = true; fail_need = 0xFFFFFFF1u; break; }      // a single record larger than the window
        uint32_t rel = (uint32_t)(off_x - stg_bit0);                          // bit cursor relative to the window
        const uint32_t pend = (uint32_t)(rec_end - stg_bit0);
        bool bad = false;
        uint64_t v;
        uint32_t d = 0;
        if (needed && lane < kwin) {                                          // readOutdegree, BVG:654-660
            const uint64_t w = win64<LIN>(stage, rel);
            const uint32_t l = GEN ? decode_generic_w(w, a.cod.outdegree, 0, &v) : gamma64(w, v);
            bad |= l == 0 || v > 0x7FFFFFFFull; rel += l; d = bad ? 0u : (uint32_t)v;
        }
        // how many leading lanes fit in the pool?
        const uint32_t dclamp = d > CAP ? CAP + 1 : d;
        const uint32_t incl = wave_incl_scan32(dclamp);
        uint32_t avail = CAP - pool_used;
        const uint32_t total = lane_get(incl, 63);
        if (!OVL && total > avail && pool_used > 0) { compact(); avail = CAP - pool_used; }
        // Scan mode stores a merged list only if a later node can copy from it (LEAN): the row is first sized
        // optimistically on the outdegrees and cut to what really fits once the references are known.
        uint32_t k = kwin;
        {
            const uint32_t budget = LEAN ? avail + (avail >> 1) : avail;
            if (total > budget) { const uint32_t kf = (uint32_t)__popcll(ballot(incl <= budget)); k = kf < k ? kf : k; }
            if (k == 0) k = 1;                                                // the exact check follows the header parse
        }
        if (needed && lane < k) nd_d[(uint32_t)x & RM] = d;
        wave_sync();

        BVG_T1(6, tq5);
        const uint32_t tq7 = BVG_T0();
        // ------------------------------------------------------------------ phase 1: parse own record
        // Steps A..D with two wave-uniform points where the scratch area (copy blocks, then intervals) is
        // allocated by prefix sums; lanes whose entries do not fit are cut from the row (k shrinks) and
        // their nodes are simply parsed again at the head of the next row.
        uint32_t ref = 0, bc = 0, ic = 0, nres = 0, sb = 0, ib = 0, ivtot = 0;
        int64_t extra = d;
        bool malf = false;                                                   // counts that contradict each other (position tasks need them exact)
        const bool parse = needed && lane < k && d > 0 && !(a.dbg & 4);
        // ---- A: reference and block count (BVG:1015-1021)
        if (parse) {
            if (W > 0) {                                                      // readReference, BVG:692-703
                const uint64_t w = win64<LIN>(stage, rel);
                uint32_t l;
                if (GEN) l = decode_generic_w(w, a.cod.reference, 0, &v);
                else { const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u; v = lz; l = lz < 64 ? lz + 1 : 0; }
                bad |= l == 0; rel += l;
                if (v > W || (int64_t)v > x) { err |= ERR_REF_RANGE; v = 0; }
                ref = (uint32_t)v;
            }
            if (ref > 0) {                                                    // readBlockCount, BVG:728-735
                const uint64_t w = win64<LIN>(stage, rel);
                const uint32_t l = GEN ? decode_generic_w(w, a.cod.block_count, 0, &v) : gamma64(w, v);
                bad |= l == 0 || v > pend - rel + 1; rel += l; bc = bad ? 0u : (uint32_t)v;
            }
        }
        const uint32_t bincl = wave_incl_scan32(bc > SCR ? SCR + 1 : bc);
        { const uint32_t kb = (uint32_t)__popcll(ballot(bincl <= SCR)); k = kb < k ? kb : k; }
        if (k == 0) { failed = true; fail_need = 0xFFFFFFF3u; break; }        // one node's copy blocks exceed the scratch area
        sb = bincl - bc;
        uint32_t btot = lane_get(bincl, k - 1);
        // ---- B: copy blocks (BVG:1023-1032) and C: interval count (BVG:1040)
        if (parse && lane < k) {
            if (ref > 0) {
                int64_t copied = 0, tot = 0;
                for (uint32_t i = 0; i < bc; i++) {
                    const uint64_t wb = win64<LIN>(stage, rel);
                    const uint32_t lb = GEN ? decode_generic_w(wb, a.cod.block, 0, &v) : gamma64(wb, v);
                    if (lb == 0 || rel > pend) { bad = true; bc = i; break; }
                    rel += lb;
                    const uint32_t b = (uint32_t)v + (i ? 1u : 0u);
                    scr[sb + i] = (T)b;
                    tot += b;
                    if (!(i & 1)) copied += b;
                }
                const int64_t rlen_ = (int64_t)nd_d[(uint32_t)(x - ref) & RM];
                if (!(bc & 1)) copied += rlen_ - tot;                         // BVG:1030
                extra = (int64_t)d - copied;
                if (tot > rlen_) malf = true;                                 // blocks running past the referenced list
                if (extra < 0 || copied < 0) { err |= ERR_MALFORMED; extra = 0; malf = true; }       // never let a tail start before the list
            }
            if (extra > 0 && minint != 0) {                                   // always gamma
                const uint32_t l = gamma64(win64<LIN>(stage, rel), v);
                bad |= l == 0 || v > (pend - rel) / 2 + 1; rel += l; ic = bad ? 0u : (uint32_t)v;
            }
        }
        const uint32_t iw = lane < k ? 2 * ic : 0u;
        const uint32_t iincl = wave_incl_scan32(iw > SCR ? SCR + 1 : iw);
        // the copy blocks of the row may have taken the whole area (interval-rich lists: transposed graphs): halve the row until the
        // first node's intervals fit behind the blocks of the nodes that stay
        for (;;) {
            const uint32_t ki = (uint32_t)__popcll(ballot(btot + iincl <= SCR));
            if (ki != 0 || k <= 1) { k = ki < k ? ki : k; break; }
            k = (k + 1u) >> 1; btot = lane_get(bincl, k - 1);
        }
        if (k == 0) { failed = true; fail_need = 0xFFFFFFF4u; break; }        // one node's intervals exceed the scratch area
        ib = btot + iincl - iw;
        // ---- D1: intervals (BVG:1042-1058): they fix the number of residuals
        if (parse && lane < k) {
            if (ic > 0) {
                int64_t prev = 0;
                for (uint32_t i = 0; i < ic; i++) {
                    uint64_t v1, v2;
                    const uint32_t l1 = gamma64(win64<LIN>(stage, rel), v1);
                    const uint32_t l2 = gamma64(win64<LIN>(stage, rel + l1), v2);
                    if (l1 == 0 || l2 == 0 || rel > pend) { bad = true; ic = i; break; }
                    rel += l1 + l2;
                    const int64_t leftv = i == 0 ? x + nat2int64(v1) : prev + 1 + (int64_t)v1;
                    const int64_t len = (int64_t)v2 + minint;
                    prev = leftv + len;
                    extra -= len; ivtot += (uint32_t)len;
                    scr[ib + 2 * i] = (T)leftv; scr[ib + 2 * i + 1] = (T)len;
                }
                if (extra < 0) { err |= ERR_MALFORMED; extra = 0; malf = true; }
            }
            nres = (uint32_t)extra;
        }
        BVG_T1(7, tq7);
        const uint32_t tq8 = BVG_T0();
        // ---- pool allocation: full list if some later node may copy it (referenced inside the row, or one of
        //      the last W nodes of the row), otherwise only the residual values
        uint64_t refmask = 0;
        if (LEAN) for (uint32_t r = 1; r <= W && r < 64; r++) refmask |= ballot(parse && lane < k && ref == r) >> r;
        uint32_t size = 0, sincl = 0, rtb = 0;
        bool stored = true;
        for (;;) {
            const uint32_t tailstart = k > W ? k - W : 0;
            stored = !LEAN || lane >= tailstart || ((refmask >> lane) & 1ull);
            uint32_t tot;
            if (TASK) {
                // lists grow from the bottom of the pool; the row's residual values are parked top-down and die
                // with the row (tasks of one node run concurrently, so they cannot share the list's own tail)
                size = (needed && lane < k && stored) ? dclamp : 0u;
                const uint32_t rsz = (needed && lane < k) ? (nres >= CAP ? CAP + 1 : nres + 1u) : 0u;     // + the guard slot of the position tasks
                sincl = wave_incl_scan32(size);
                const uint32_t rincl = wave_incl_scan32(rsz);
                rtb = CAP - (rincl > CAP ? CAP : rincl);
                tot = sincl + rincl;
                if (OVL && rincl + stage_off + SWE > CAP) tot = 0xFFFFFFFFu;     // the parked residuals may not reach down into the window
            } else {
                size = (needed && lane < k) ? (stored ? dclamp : (nres > CAP ? CAP + 1 : nres)) : 0u;
                sincl = wave_incl_scan32(size);
                tot = sincl;
            }
            if (lane_get(tot, k - 1) <= avail) break;
            const uint32_t kf = (uint32_t)__popcll(ballot(tot <= avail && lane < k));
            if (kf == 0) { k = 0; break; }
            k = kf;
        }
        if (k == 0) {                                                         // first node alone overflows the pool
            failed = true;
            {
                uint32_t d0 = lane_get(d, 0); const uint32_t n0 = lane_get(nres, 0);
                if (TASK && d0 <= 0x3FFFFFFFu) d0 += (n0 > d0 ? d0 : n0) + 1u;
                fail_need = d0 > 0x3FFFFFFFu ? 0xFFFFFFF2u : d0 + pool_used + (d0 >> 2) + 64 + SWE;
            }
            break;
        }
        const bool act = needed && lane < k;
        const uint32_t base = pool_used + (sincl - size);
        if (act) nd_base[(uint32_t)x & RM] = base;
        pool_used += lane_get(sincl, k - 1);
        if (OVL && pool_used > stage_off) ovl_dirty = true;
        // prefetch the next row's offsets (their latency hides behind the rest of this row's decode)
        uint64_t nxt_off = 0, nxt_end = 0;
        {
            const int64_t nx = r0 + k + lane;
            if (nx < e) { nxt_off = a.offsets[nx]; nxt_end = a.offsets[nx + 1]; }
        }
        BVG_T1(8, tq8);
        const uint32_t tq9 = BVG_T0();
        // ---- D2: residuals (ResidualLongIterator, BVG:902-935) to the node's parking area in the pool
        const uint32_t recrel = (uint32_t)(off_x - stg_bit0);
        uint32_t cntE = 0, efirst = 0;
        if (sk_track) {                                                       // skip entries of the row, in node order
            cntE = (parse && lane < k && !bad && nres >= kSkipMin) ? (nres - 1u) >> kSkipShift : 0u;
            const uint32_t eincl = wave_incl_scan32(cntE);
            efirst = sk_run + eincl - cntE;
            sk_run += lane_get(eincl, 63);
        }
        const uint32_t rdst = TASK ? rtb : base + size - nres;
        if ((a.skip_mode == 0 || a.skip_mode == 3) && sk_n != 0 && ballot(cntE != 0) && !(a.dbg & 2)) {   // (3: the validating pass of the index build decodes WITH the entries the dense walk has just written, and checks every one of them)
            // long residual lists are cut at their skip entries: every segment of <= kSkipEvery gaps is one task
            if (sk_run > sk_n) { failed = true; fail_need = 0xFFFFFFF5u; break; }      // index out of step with the stream
            // long tasks first (full segments and tails of more than kShortTask gaps), the short tails after them: a pass of 64
            // tasks lasts as long as its longest one, so like goes with like
            const bool hasres = parse && lane < k && nres > 0 && !bad;
            const uint32_t lastc = nres - (cntE << kSkipShift);
            const bool shortt = hasres && lastc <= kShortTask;
            const uint32_t Tn = hasres ? cntE + (shortt ? 0u : 1u) : 0u;           // long tasks of this node
            const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, NL = lane_get(tincl, 63);
            const uint64_t smask = ballot(shortt);
            const uint32_t ss = NL + (uint32_t)__popcll(smask & ((1ull << lane) - 1ull)), Ttot = NL + (uint32_t)__popcll(smask);
            bool tbad = false;
            // RU tasks per lane and pass, decoded in one interleaved loop: a gap is a chain of dependent LDS reads and shifts
            // (~300 cycles), and with two wavefronts per SIMD nothing else hides it -- two independent chains per lane do.
            auto task_passes = [&](auto RUc) {
            constexpr uint32_t RU = decltype(RUc)::value, RP = 64u * RU;
            for (uint32_t p0 = 0; p0 < Ttot; p0 += RP) {
                {
                    const uint32_t q0 = ts < p0 ? p0 - ts : 0u;
                    const uint32_t q1 = ts >= p0 + RP ? 0u : (ts + Tn > p0 + RP ? p0 + RP - ts : Tn);
                    for (uint32_t q = q0; q < q1; q++) rtmap[ts + q - p0] = lane | (q << 8);
                    if (shortt && ss >= p0 && ss < p0 + RP) rtmap[ss - p0] = lane | (cntE << 8);
                }
                wave_sync();
                bool tl[RU]; uint32_t cnt[RU], trel[RU], tpend[RU], tfirst[RU], taddr[RU], tlast[RU], tchk[RU], trec[RU]; T r[RU];
#pragma unroll
                for (uint32_t u = 0; u < RU; u++) {
                    tl[u] = p0 + 64u * u + lane < Ttot;
                    const uint32_t ent = tl[u] ? rtmap[64u * u + lane] : lane;
                    const int nl = (int)(ent & 63u); const uint32_t q = ent >> 8;
                    const uint32_t t_rel = __shfl(rel, nl, 64), t_rec = __shfl(recrel, nl, 64), t_pend = __shfl(pend, nl, 64);
                    const uint32_t t_nres = __shfl(nres, nl, 64), t_dst = __shfl(rdst, nl, 64), t_ef = __shfl(efirst, nl, 64);
                    const uint32_t t0 = q << kSkipShift;
                    const uint32_t t_ce = t_nres >= kSkipMin ? (t_nres - 1u) >> kSkipShift : 0u;      // the node's entries
                    cnt[u] = tl[u] ? (q == t_ce ? t_nres - t0 : kSkipEvery) : 0u;             // the last segment takes the remainder
                    trel[u] = tl[u] ? t_rel : 0u; r[u] = (T)(r0 + nl); tpend[u] = t_pend;
                    tfirst[u] = (tl[u] && q == 0) ? 1u : 0u;
                    tlast[u] = (tl[u] && t0 + cnt[u] == t_nres) ? 1u : 0u;
                    taddr[u] = t_dst + t0;
                    // skip_mode 3: a segment that is not its list's last must END exactly where the next entry says the next one starts, on the value it holds:
                    // segment 0 starts where the header parse ended, so by induction every entry of a block that passes is the true walk's (bvg_index.hip)
                    trec[u] = t_rec; tchk[u] = (a.skip_mode == 3 && tl[u] && q < t_ce) ? t_ef + q + 1u : 0u;
                    if (tl[u] && q) {
                        const uint64_t e = sk_base + t_ef + q - 1u;
                        trel[u] = t_rec + a.skip_bit[e]; r[u] = reinterpret_cast<const T*>(a.skip_val)[e];
                        if (!(trel[u] > t_rel && trel[u] < t_pend) || trel[u] - t_rec == 0xFFFFu) { tbad = true; cnt[u] = 0; trel[u] = 0; }
                    }
                }
                auto decode_tasks = [&](auto ZF) {                            // ZF: the 32-bit zeta fast path is compiled in (no test inside the loop)
                    for (uint32_t i = 0;; i++) {
                        bool on[RU]; bool any = false;
#pragma unroll
                        for (uint32_t u = 0; u < RU; u++) { on[u] = i < cnt[u]; any |= on[u]; }
                        if (!ballot(any)) break;
                        uint32_t len[RU]; uint64_t val[RU]; bool slow = false;
                        uint32_t w32[RU];
                        if (decltype(ZF)::value) {
#pragma unroll
                            for (uint32_t u = 0; u < RU; u++) w32[u] = win32<LIN>(stage, trel[u]);   // all chains' LDS reads first: they overlap
                        }
#pragma unroll
                        for (uint32_t u = 0; u < RU; u++) {
                            len[u] = 0; val[u] = 0;
                            if (decltype(ZF)::value) { uint32_t v32; len[u] = zeta_fast32(w32[u], zk, v32); val[u] = v32; }
                            slow |= on[u] && len[u] == 0;
                        }
                        if (ballot(slow)) {                                   // codes longer than 31 bits / other codings: one rare, wave-uniform detour
#pragma unroll
                            for (uint32_t u = 0; u < RU; u++)
                                if (on[u] && len[u] == 0) {
                                    const uint64_t w = win64<LIN>(stage, trel[u]);
                                    len[u] = GEN ? decode_generic_w(w, a.cod.residual, zk, &val[u]) : zeta64(w, zk, val[u]);
                                    if (len[u] == 0) { tbad = true; cnt[u] = 0; on[u] = false; }
                                }
                        }
#pragma unroll
                        for (uint32_t u = 0; u < RU; u++) {
                            const T gap = (tfirst[u] && i == 0) ? (T)nat2int64(val[u]) : (T)(1 + (T)val[u]);
                            const T rn = (T)(r[u] + gap);
                            const uint32_t tn = trel[u] + len[u];
                            if (on[u]) pool[taddr[u] + i] = rn;
                            r[u] = on[u] ? rn : r[u]; trel[u] = on[u] ? tn : trel[u];
                            const bool over = on[u] && tn > tpend[u];
                            err |= over ? ERR_OVERRUN : 0u; cnt[u] = over ? 0u : cnt[u]; tlast[u] = over ? 0u : tlast[u];
                        }
                    }
                };
                if (zfast) decode_tasks(std::true_type{}); else decode_tasks(std::false_type{});
#pragma unroll
                for (uint32_t u = 0; u < RU; u++)
                    if (tlast[u] && cnt[u] && trel[u] != tpend[u] && !tbad && !(a.dbg & 7u)) err |= ERR_MALFORMED;
#pragma unroll
                for (uint32_t u = 0; u < RU; u++)
                    if (tchk[u] && cnt[u]) {
                        const uint64_t en = sk_base + tchk[u] - 1u;
                        if (trel[u] != trec[u] + a.skip_bit[en] || r[u] != reinterpret_cast<const T*>(a.skip_val)[en]) tbad = true;
                    }
                wave_sync();
            }
            };
            // two chains per lane only when the row has the tasks to fill them: a half-empty second chain doubles the instructions of
            // every step for nothing (reference-free graphs: 166 G edges/s with one chain, 136 G with two, profiles/r02)
            if (kResUnroll >= 2 && Ttot > 96u) task_passes(std::integral_constant<uint32_t, 2>{}); else task_passes(std::integral_constant<uint32_t, 1>{});
            if (((act && d == 0) || (parse && lane < k && nres == 0)) && rel != pend && !bad && !(a.dbg & 7u)) err |= ERR_MALFORMED;
            bad |= tbad;
            if (ballot(tbad)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
        } else if (parse && lane < k) {
            if (nres > 0 && !bad && !(a.dbg & 2)) {
                T* const tail = pool + rdst;
                T r = (T)x;
                for (uint32_t t = 0; t < nres; t++) {
                    if (a.skip_mode == 2 && cntE && t && (t & (kSkipEvery - 1u)) == 0) {  // fill the skip entry of this residual
                        const uint32_t ei = efirst + (t >> kSkipShift) - 1u;                // inside the block's allotment only: a block
                        if (ei < sk_n) {                                                    // that ends in the generic kernel has none
                            a.skip_bit[sk_base + ei] = (uint16_t)(rel - recrel < 0xFFFFu ? rel - recrel : 0xFFFFu); reinterpret_cast<T*>(a.skip_val)[sk_base + ei] = r;   // (0xFFFF: unusable, the reader fails over)
                        }
                    }
                    uint64_t val;
                    const uint32_t len = read_residual<GEN>(stage, rel, zfast, zk, a.cod.residual, val);
                    if (len == 0) { bad = true; break; }
                    rel += len;
                    r = t == 0 ? (T)(r + (T)nat2int64(val)) : (T)(r + 1 + (T)val);
                    tail[t] = r;
                    if (rel > pend) { err |= ERR_OVERRUN; break; }
                }
            }
            if (rel != pend && !bad && !(a.dbg & 7u)) err |= ERR_MALFORMED;         // SURVEY A.6 self-check
        } else if (act && d == 0 && rel != pend && !bad) err |= ERR_MALFORMED;
        if (ballot(bad && lane < k)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
        wave_sync();

        // ------------------------------------------------------------------ phase 2: data-flow emission
        const bool rep = act && x >= rep_lo && x < rep_hi;
        uint32_t k0 = 0, k1 = 0;
        if (rep && !MAT) {
            node_key((uint64_t)x + a.node_base, k0, k1);
        }
        BVG_T1(9, tq9);
        BVG_T1(5, tq5);
        bool by_tasks = false;
        if constexpr (TASK) {
            // Level-synchronous emission by POSITION.  Nodes are grouped by their depth in the row's reference forest.  The
            // successor list of a node is the masked copy of the referenced list with the extras (residuals, intervals)
            // inserted (BVG:1062-1090), and the three streams are disjoint in a well-formed file, so every extra knows its
            // place without a merge: (extras below it) + (copied elements below it).  Per level:
            //   Z1  one lane per EXTRA (residual value or interval): lower bound in the referenced list + rank under the copy
            //       mask give its output position; residual values are stored (and summed) right there, positions are kept;
            //   Z2  one lane per TASK of S consecutive output positions, all tasks of a level equally long: a position is
            //       a residual (done), falls into an interval, or takes the next kept element of the referenced list.
            // An extra that meets a copied element or another extra (which the reference's merge would emit once,
            // MergedLongIterator.java:85-89) sends the block to the generic kernel, which follows the iterators literally.
            uint32_t* const tmap = produced;
            constexpr uint32_t HS = sizeof(T) * 4;                            // interval entry: length | position << HS
            const T HM = (T)(((T)1 << HS) - 1);
            const uint32_t tq0 = BVG_T0();
            uint32_t rlbN = 0, rlenN = 0;
            if (act && ref > 0) { const int64_t y = x - ref; rlbN = nd_base[(uint32_t)y & RM]; rlenN = nd_d[(uint32_t)y & RM]; }
            // A node without reference is its residuals merged with its intervals (BVG:1087-1089): the parked residual
            // values play the role of an unmasked "referenced list", the intervals are the only extras to place (and the
            // guard serves as its empty array of residual positions).
            const bool pure = act && ref == 0;
            if (pure) { rlbN = rtb; rlenN = nres; }
            const uint32_t rtbN = pure ? rtb + nres : rtb, nresN = pure ? 0u : nres;
            const bool inrow = act && ref > 0 && ref <= lane;
            uint32_t lvl = 0;
            for (int it = 0; it < 64; it++) {
                const uint32_t up = __shfl(lvl, inrow ? (int)(lane - ref) : (int)lane, 64);
                const uint32_t nl = inrow ? up + 1 : 0;
                const bool ch = nl != lvl; lvl = nl;
                if (!ballot(ch)) break;
            }
            const bool emitn = act && d > 0 && !(a.dbg & 1);
            // Position tasks pay a fixed cost per level (maps, seeks); the pipelined node-per-lane loop below costs about the
            // longest list of the row.  Estimate both and take the cheaper one for this row.
            const uint32_t rowW = wave_sum32(emitn ? d : 0u);
            {
                const uint32_t maxd = wave_max32(emitn ? d : 0u), nlev = wave_max32(emitn ? lvl + 1u : 0u);
                const uint32_t est = ((rowW * 21u) >> 10) + nlev * a.pass_cost;
                by_tasks = (est < maxd || (a.dbg & 16u) || a.skip_mode >= 2) && !(a.dbg & 32u);   // (the index-filling pass is the validating pass: bvg_scan.hip)
            }
            bool zbad = malf;
            if (by_tasks && act && bc) {                                      // copy blocks -> prefix form (MaskPrefix), once per row
                uint32_t pp = 0, kk = 0;
                for (uint32_t i = 0; i < bc; i++) {
                    const uint32_t b = (uint32_t)scr[sb + i];
                    pp += b; if (!(i & 1u)) kk += b;
                    scr[sb + i] = MaskPrefix<T>::pack(pp, kk);
                }
            }
            uint64_t remaining = by_tasks ? ballot(emitn) : 0ull;
            if (by_tasks && act) pool[rtb + nres] = sentinel<T>();            // guard behind the node's residual positions
            BVG_T1(0, tq0);
            for (uint32_t L = 0; remaining; L++) {
                const uint32_t tq1 = BVG_T0();
                const bool mem = emitn && lvl == L;
                remaining &= ~ballot(mem);
                // ---------------- Z1: one lane per extra
                {
                    const uint32_t In = mem ? nresN + ic : 0u;
                    const uint32_t iincl2 = wave_incl_scan32(In), is = iincl2 - In, Itot = lane_get(iincl2, 63);
                    for (uint32_t p0 = 0; p0 < Itot; p0 += 64) {
                        {
                            const uint32_t q0 = is < p0 ? p0 - is : 0u;
                            const uint32_t q1 = is >= p0 + 64u ? 0u : (is + In > p0 + 64u ? p0 + 64u - is : In);
                            for (uint32_t q = q0; q < q1; q++) tmap[is + q - p0] = lane | (q << 8);
                        }
                        wave_sync();
                        const bool tl = p0 + lane < Itot;
                        const uint32_t ent = tl ? tmap[lane] : lane;
                        const int nl = (int)(ent & 63u); const uint32_t q = ent >> 8;
                        const uint32_t t_d = __shfl(d, nl, 64), t_rlb = __shfl(rlbN, nl, 64), t_rlen = __shfl(rlenN, nl, 64);
                        const uint32_t t_bc = __shfl(bc, nl, 64), t_sb = __shfl(sb, nl, 64), t_ic = __shfl(ic, nl, 64), t_ib = __shfl(ib, nl, 64);
                        const uint32_t t_nres = __shfl(nresN, nl, 64), t_rtb = __shfl(rtbN, nl, 64), t_ob = __shfl(base, nl, 64);
                        const uint32_t t_fl = __shfl((uint32_t)stored | ((uint32_t)rep << 1), nl, 64);
                        const uint32_t t_k1 = __shfl(k1, nl, 64);
                        const T* const rl = pool + t_rlb; T* const rt = pool + t_rtb;
                        T v = 0; uint32_t len = 1, pe = 0; bool isiv = false;
                        if (tl) {
                            uint32_t eb;
                            if (q < t_ic) {                                   // interval q: the intervals and residuals below it
                                isiv = true;
                                v = scr[t_ib + 2 * q]; len = (uint32_t)(scr[t_ib + 2 * q + 1] & HM);
                                eb = 0;
                                for (uint32_t i = 0; i < q; i++) eb += (uint32_t)(scr[t_ib + 2 * i + 1] & HM);
                                const uint32_t lb = lds_lower_bound<T>(rt, t_nres, v);
                                if (lb < t_nres && (T)(rt[lb] - v) < (T)len) zbad = true;          // a residual inside the interval
                                eb += lb;
                            } else {                                          // residual q - ic
                                const uint32_t i = q - t_ic;
                                v = rt[i]; eb = i;
                                for (uint32_t kk = 0; kk < t_ic; kk++) {
                                    const T left = scr[t_ib + 2 * kk]; const uint32_t ln = (uint32_t)(scr[t_ib + 2 * kk + 1] & HM);
                                    if (left <= v) { eb += ln; if ((T)(v - left) < (T)ln) zbad = true; }
                                }
                            }
                            uint32_t t = 0;
                            if (t_rlen) {                                     // copied elements below v: rank of its lower bound under the mask
                                const uint32_t qq = lds_lower_bound<T>(rl, t_rlen, v);
                                uint32_t qn;
                                t = MaskPrefix<T>::rank(scr + t_sb, t_bc, t_rlen, qq, qn);
                                if (qn < t_rlen && (T)(rl[qn] - v) < (T)len) zbad = true;           // a copied element meets the extra
                            }
                            pe = eb + t;
                            if (pe + len > t_d) { zbad = true; pe = 0; len = 0; }
                        }
                        wave_sync();                                      // the parked values have been read: positions may replace them
                        if (tl && len) {
                            if (isiv) scr[t_ib + 2 * q + 1] = (T)len | (T)((T)pe << HS);
                            else {
                                if (t_fl & 1u) pool[t_ob + pe] = v;
                                if (!MAT && (t_fl & 2u)) blk_chk += mix_node<T>(t_k1, v);
                                rt[q - t_ic] = (T)pe;
                            }
                        }
                        cnt_seek++;
                        wave_sync();
                    }
                }
                BVG_T1(3, tq1);
                const uint32_t tq1b = BVG_T0();
                // ---------------- Z2: tasks of S output positions (lists that some later node may copy; the others wait for the leaf pass)
                const bool mem2 = mem && stored;
                const uint32_t Wl = wave_sum32(mem2 ? d : 0u);
                uint32_t S = (Wl + 63u) >> 6; if (S < kMinTask) S = kMinTask;
                uint32_t Tn = 0;
                for (int it = 0; it < 6; it++) {
                    Tn = 0;
                    if (mem2) { Tn = (uint32_t)((float)d / (float)S); while (Tn * S < d) Tn++; while (Tn > 1u && (Tn - 1u) * S >= d) Tn--; }
                    const uint32_t tt = wave_sum32(Tn);
                    if (tt <= 64u || it == 5) break;
                    const uint32_t s2 = (uint32_t)((float)S * (float)tt * (1.0f / 64.0f));
                    S = s2 > S ? s2 : S + 1u;
                    }
                const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, Ttot = lane_get(tincl, 63);
                BVG_T1(1, tq1b);
                for (uint32_t p0 = 0; p0 < Ttot; p0 += 64) {
                    const uint32_t tq2 = BVG_T0();
                    {   // task map of this pass: (node lane, task index inside the node)
                        const uint32_t q0 = ts < p0 ? p0 - ts : 0u;
                        const uint32_t q1 = ts >= p0 + 64u ? 0u : (ts + Tn > p0 + 64u ? p0 + 64u - ts : Tn);
                        for (uint32_t q = q0; q < q1; q++) tmap[ts + q - p0] = lane | (q << 8);
                    }
                    wave_sync();
                    const bool tl = p0 + lane < Ttot;
                    const uint32_t ent = tl ? tmap[lane] : lane;
                    const int nl = (int)(ent & 63u); const uint32_t q = ent >> 8;
                    const uint32_t t_d = __shfl(d, nl, 64), t_rlb = __shfl(rlbN, nl, 64), t_rlen = __shfl(rlenN, nl, 64);
                    const uint32_t t_bc = __shfl(bc, nl, 64), t_sb = __shfl(sb, nl, 64), t_ic = __shfl(ic, nl, 64), t_ib = __shfl(ib, nl, 64);
                    const uint32_t t_nres = __shfl(nresN, nl, 64), t_rtb = __shfl(rtbN, nl, 64), t_ob = __shfl(base, nl, 64);
                    const uint32_t t_fl = __shfl((uint32_t)stored | ((uint32_t)rep << 1), nl, 64);
                    const uint32_t t_k1 = __shfl(k1, nl, 64);
                    const bool t_stored = t_fl & 1u, t_rep = (t_fl >> 1) & 1u;
                    const T* const rl = pool + t_rlb; const T* const rt = pool + t_rtb; T* const out = pool + t_ob;
                    uint32_t p = 0, pstop = 0, ri = 0, rnext = kInf, ivk = t_ic, ivpos = kInf, ivlen = 0, qcur = 0, krem = kInf, bi = t_bc;
                    T ivleft = 0;
                    if (tl) {
                        p = q * S; pstop = p + S < t_d ? p + S : t_d;
                        ri = lds_lower_bound<T>(rt, t_nres, (T)p);            // residual positions below p
                        rnext = (uint32_t)rt[ri];                             // (the guard reads as kInf)
                        uint32_t ie = 0;
                        for (uint32_t i = 0; i < t_ic; i++) {                 // interval elements below p; the interval at / after p
                            const T pk = scr[t_ib + 2 * i + 1];
                            const uint32_t ln = (uint32_t)(pk & HM), ps = (uint32_t)(pk >> HS);
                            if (ps + ln > p) { ivk = i; ivpos = ps; ivlen = ln; ivleft = scr[t_ib + 2 * i]; if (p > ps) ie += p - ps; break; }
                            ie += ln;
                        }
                        const uint32_t t = p - ri - ie;                       // rank of the next copied element among the kept ones
                        if (t_rlen) MaskPrefix<T>::select(scr + t_sb, t_bc, t_rlen, t, qcur, krem, bi);   // MaskedLongIterator.java:73-100: the t-th kept position
                    }
                    BVG_T1(2, tq2);
                    const uint32_t tq4 = BVG_T0();
                    cnt_pass++; cnt_tasks += (uint32_t)__popcll(ballot(tl));
                    if (a.dbg & 128u) pstop = p;
                    const uint32_t rlast = t_rlen ? t_rlen - 1u : 0u;
                    for (;;) {
                        const bool todo = p < pstop;
                        if (!ballot(todo)) break;
                        cnt_iter++;
                        if (todo) {
                            if (p == rnext) { ri++; rnext = (uint32_t)rt[ri]; }           // a residual: placed by Z1
                            else {
                                const uint32_t io = p - ivpos;
                                const bool ii = io < ivlen;                               // LongIntervalSequenceIterator.java:71-78
                                const T cv = rl[qcur < rlast ? qcur : rlast];
                                const T v = ii ? (T)(ivleft + (T)io) : cv;
                                if (t_stored) out[p] = v;
                                if (!MAT && t_rep) blk_chk += mix_node<T>(t_k1, v);
                                if (ii) {
                                    if (io + 1u == ivlen) {
                                        ivk++; ivpos = kInf; ivlen = 0;
                                        if (ivk < t_ic) { const T pk = scr[t_ib + 2 * ivk + 1]; ivlen = (uint32_t)(pk & HM); ivpos = (uint32_t)(pk >> HS); ivleft = scr[t_ib + 2 * ivk]; }
                                    }
                                } else {
                                    qcur++;
                                    if (--krem == 0) MaskPrefix<T>::next_block(scr + t_sb, t_bc, t_rlen, qcur, krem, bi);   // MaskedLongIterator.java:81-100
                                }
                            }
                            p++;
                        }
                    }
                    wave_sync();
                    BVG_T1(4, tq4);
                }
            }
            // ---------------- leaf pass (scan mode): lists that no later node copies are never materialised, so they need no
            // output positions at all -- their residuals were summed by Z1, and what is left is to sum the kept elements of the
            // referenced list (MaskedLongIterator.java:73-100) and the interval elements, in equal tasks of S elements over ALL
            // such nodes of the row at once (every referenced list is complete by now).
            const uint32_t tqL = BVG_T0();
            if (LEAN && by_tasks && ballot(emitn && !stored)) {
                const bool lf = emitn && !stored && !zbad;
                const uint32_t keptN = lf ? (pure ? nres : d - nres - ivtot) : 0u;      // copied elements (the parked residuals of a reference-free node)
                const uint32_t ivN = lf ? ivtot : 0u;
                const uint32_t Wk = wave_sum32(keptN + ivN);
                uint32_t S = (Wk + 63u) >> 6; if (S < kMinTask) S = kMinTask;
                uint32_t Tk = 0, Tn = 0;
                for (int it = 0; it < 6; it++) {                               // the smallest S for which the tasks fit one pass
                    Tk = (keptN + S - 1u) / S; Tn = Tk + (ivN + S - 1u) / S;
                    const uint32_t tt = wave_sum32(Tn);
                    if (tt <= 64u || it == 5) break;
                    const uint32_t s2 = (uint32_t)((float)S * (float)tt * (1.0f / 64.0f));
                    S = s2 > S ? s2 : S + 1u;
                }
                const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, Ttot = lane_get(tincl, 63);
                for (uint32_t p0 = 0; p0 < Ttot; p0 += 64) {
                    {
                        const uint32_t q0 = ts < p0 ? p0 - ts : 0u;
                        const uint32_t q1 = ts >= p0 + 64u ? 0u : (ts + Tn > p0 + 64u ? p0 + 64u - ts : Tn);
                        for (uint32_t q = q0; q < q1; q++) tmap[ts + q - p0] = lane | (q << 8);
                    }
                    wave_sync();
                    const bool tl = p0 + lane < Ttot;
                    const uint32_t ent = tl ? tmap[lane] : lane;
                    const int nl = (int)(ent & 63u); const uint32_t q = ent >> 8;
                    const uint32_t t_rlb = __shfl(rlbN, nl, 64), t_rlen = __shfl(rlenN, nl, 64), t_bc = __shfl(bc, nl, 64), t_sb = __shfl(sb, nl, 64);
                    const uint32_t t_ic = __shfl(ic, nl, 64), t_ib = __shfl(ib, nl, 64), t_kept = __shfl(keptN, nl, 64), t_tk = __shfl(Tk, nl, 64), t_iv = __shfl(ivN, nl, 64);
                    const uint32_t t_k1r = __shfl(rep ? k1 : 0u, nl, 64);     // a node outside [from,to) sums nothing
                    const T* const rl = pool + t_rlb;
                    uint32_t cnt = 0, qcur = 0, krem = kInf, bi = t_bc, ivk = 0, ivrem = 0; bool iota = false; T ivv = 0;
                    if (tl) {
                        if (q < t_tk) {                                        // S kept elements from the t-th on
                            const uint32_t t = q * S;
                            cnt = t_kept - t < S ? t_kept - t : S;
                            MaskPrefix<T>::select(scr + t_sb, t_bc, t_rlen, t, qcur, krem, bi);
                        } else {                                               // S elements of the node's intervals, taken as one sequence
                            uint32_t e0 = (q - t_tk) * S; iota = true;        // (LongIntervalSequenceIterator.java:71-78)
                            cnt = t_iv - e0 < S ? t_iv - e0 : S;
                            for (; ivk < t_ic; ivk++) {
                                const uint32_t ln = (uint32_t)(scr[t_ib + 2 * ivk + 1] & HM);
                                if (e0 < ln) { ivv = (T)(scr[t_ib + 2 * ivk] + (T)e0); ivrem = ln - e0; break; }
                                e0 -= ln;
                            }
                        }
                    }
                    const uint32_t rlast = t_rlen ? t_rlen - 1u : 0u;
                    uint64_t csum = 0;
                    const uint32_t tqL2 = BVG_T0();
                    cnt_leafp++;
                    for (uint32_t i = 0;; i++) {
                        const bool todo = i < cnt;
                        if (!ballot(todo)) break;
                        cnt_leaf++;
                        const T cv = rl[qcur < rlast ? qcur : rlast];
                        const T v = iota ? ivv : cv;
                        csum += mix_node<T>(todo ? t_k1r : 0u, v);
                        if (todo) {
                            if (iota) {
                                ivv++;
                                if (--ivrem == 0 && ++ivk < t_ic) { ivv = scr[t_ib + 2 * ivk]; ivrem = (uint32_t)(scr[t_ib + 2 * ivk + 1] & HM); }
                            } else {
                                qcur++;
                                if (--krem == 0) MaskPrefix<T>::next_block(scr + t_sb, t_bc, t_rlen, qcur, krem, bi);   // MaskedLongIterator.java:81-100
                            }
                        }
                    }
                    blk_chk += csum;
                    BVG_T1(11, tqL2);
                    wave_sync();
                }
            }
            BVG_T1(10, tqL);
            if (by_tasks) {
                if (ballot(zbad)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
                if (rep) { blk_arcs += d; blk_nodes += 1; if (!MAT) blk_chk += mix_node_const(k0, k1, a.node_base, d); }
            }
        }
        if (!by_tasks) {
            validated = false;
            produced[lane] = act ? 0u : kInf;
            wave_sync();
            T* const out = pool + base;
            const T* rl = pool; uint32_t rlen = 0, rpos = 0, keep = 0, bi = 0; uint32_t rlane = lane; bool samerow = false;
            if (act && ref > 0) {
                const int64_t y = x - ref;
                rl = pool + nd_base[(uint32_t)y & RM]; rlen = nd_d[(uint32_t)y & RM];
                if (y >= r0) { rlane = lane - ref; samerow = true; }
                if (bc == 0) keep = kInf;                                         // MaskedLongIterator.java:73-78
                else {
                    keep = (uint32_t)scr[sb]; bi = 1;
                    if (keep == 0) {
                        if (bi >= bc) rpos = rlen;
                        else { rpos += (uint32_t)scr[sb + bi]; bi++; if (bi >= bc) keep = kInf; else { keep = (uint32_t)scr[sb + bi]; bi++; } }
                    }
                }
            }
            T ivcur = 0; uint32_t ivrem = 0, ivi = 0;
            if (act && ic > 0) { ivcur = scr[ib]; ivrem = (uint32_t)scr[ib + 1]; ivi = 1; }      // (act: a lane behind the row's cut has an interval count but no place in the scratch area -- tests/emu under AddressSanitizer)
            uint32_t rsi = 0;
            const T* const rtail = pool + rdst;                                    // the node's residual values
            T rhead = (act && nres) ? rtail[0] : sentinel<T>();                    // (act: a lane behind the row's cut has a residual count but no list: its `rdst` wraps)
            uint32_t j = 0;
            uint64_t chk = 0;
            for (;;) {
                const bool todo = act && j < d && !(a.dbg & 1);
                if (!ballot(todo)) break;
                cnt_iter++;
                const bool cneed = todo && rpos < rlen;
                // both loads are issued together; the copy head is only used when the producer is far enough
                const uint32_t pr = __hip_atomic_load(&produced[rlane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const T cval = rl[cneed ? rpos : 0];
                const bool cready = !cneed || !samerow || pr > rpos;
                if (todo && cready) {
                    const T c = cneed ? cval : sentinel<T>();
                    const T iv = ivrem ? ivcur : sentinel<T>();
                    T m = c < iv ? c : iv; m = m < rhead ? m : rhead;             // MergedLongIterator.java:63-92, three-way
                    if (stored) out[j] = m;
                    j++;
                    __hip_atomic_store(&produced[lane], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (!MAT && rep) chk += m == sentinel<T>() ? (uint64_t)k1 * (~0ull - a.node_base) : mix_node<T>(k1, m);   // (-1 where the merge ran dry, BVG:1164-1176: the per-node constant adds the base to every arc)
                    if (cneed && c == m) {                                        // MaskedLongIterator.java:81-100
                        rpos++;
                        if (--keep == 0) {
                            if (bi >= bc) rpos = rlen;
                            else { rpos += (uint32_t)scr[sb + bi]; bi++; if (bi >= bc) keep = kInf; else { keep = (uint32_t)scr[sb + bi]; bi++; } }
                        }
                    }
                    if (ivrem && iv == m) {                                       // LongIntervalSequenceIterator.java:71-78
                        ivcur++;
                        if (--ivrem == 0 && ivi < ic) { ivcur = scr[ib + 2 * ivi]; ivrem = (uint32_t)scr[ib + 2 * ivi + 1]; ivi++; }
                    }
                    if (rsi < nres && rhead == m) { rsi++; rhead = rsi < nres ? rtail[rsi] : sentinel<T>(); }
                }
            }
            if (rep) { blk_arcs += d; blk_chk += chk; blk_nodes += 1; if (!MAT) blk_chk += mix_node_const(k0, k1, a.node_base, d); }
        }

        // ------------------------------------------------------------------ materialise: coalesced copy-out
        if (MAT) {
            wave_sync();
            const uint64_t repmask = ballot(rep);
            if (repmask) {
                const int la = __ffsll((unsigned long long)repmask) - 1;
                const int lb = 63 - __clzll(repmask);
                const uint32_t seg0 = lane_get(base, (uint32_t)la);
                const uint32_t seg1 = lane_get(base + d, (uint32_t)lb);
                const uint64_t dst0 = a.batch ? a.cum[bid >> 1] : a.cum[(r0 + la) - a.from];
                for (uint32_t t = lane; t < seg1 - seg0; t += 64) {
                    const T vv = pool[seg0 + t];
                    a.succ[dst0 + t] = vv == sentinel<T>() ? -1ll : (int64_t)((uint64_t)vv + a.node_base);
                }
                if (rep && a.outdeg && !a.batch) a.outdeg[x - a.from] = (int32_t)d;
            }
        }
        wave_sync();
        // next row: lanes shift by k; reuse the prefetched offsets
        cnt_rows++;
        r0 += k;
        off_x = nxt_off; rec_end = nxt_end;
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
        if (a.skip_mode == 1 && a.skip_cnt) a.skip_cnt[bid] = sk_run;
        if (a.skip_mode >= 2 && a.skip_fmt && sk_have) a.skip_fmt[bid] = (TASK && validated && err == 0) ? 1 : 3;   // 1: the lean scan kernel may take the block
        unsigned long long* const accs = a.acc + (size_t)(bid & a.acc_mask) * kAccStride;   // this block's result stripe
        atomicAdd(&accs[0], (unsigned long long)blk_arcs);
        atomicAdd(&accs[1], (unsigned long long)blk_chk);
        atomicAdd(&accs[2], (unsigned long long)blk_nodes);
        if (err) atomicOr(&accs[3], (unsigned long long)err);
        if (a.dbg & 64u) {
            atomicAdd(&a.acc[4], (unsigned long long)cnt_iter); atomicAdd(&a.acc[5], (unsigned long long)cnt_pass); atomicAdd(&a.acc[6], (unsigned long long)cnt_rows);
            atomicAdd(&a.acc[7], (unsigned long long)cnt_tasks); atomicAdd(&a.acc[8], (unsigned long long)cnt_seek);
            atomicAdd(&a.acc[22], (unsigned long long)cnt_leaf); atomicAdd(&a.acc[23], (unsigned long long)cnt_leafp);
#ifdef BVG_PROF
            for (int i = 0; i < 10; i++) atomicAdd(&a.acc[9 + i], (unsigned long long)cyc[i]);
            atomicAdd(&a.acc[20], (unsigned long long)cyc[10]); atomicAdd(&a.acc[21], (unsigned long long)cyc[11]);
#endif
        }
    }
}

}  // namespace

void launch_rows_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, bool materialise, hipStream_t s) {
    if (nblocks == 0) return;
    dim3 grid(nblocks), block(64);
    const bool gen = !(a.cod.outdegree == BVG_GAMMA && a.cod.reference == BVG_UNARY && a.cod.block_count == BVG_GAMMA &&
                       a.cod.block == BVG_GAMMA && a.cod.residual == BVG_ZETA);
    const bool task0 = a.emit_tasks != 0;
    size_t dyn = (size_t)(a.lds_pool_elems + a.lds_scr_elems) * (wide ? 8 : 4) + (task0 ? 0 : (size_t)a.lds_stage_words * 4);   // (task variant: the window lies inside the pool)
    if (knob("BVG_LDSPAD")) dyn += (size_t)atoi(knob("BVG_LDSPAD"));   // occupancy experiments: unused LDS behind the window
    const bool task = a.emit_tasks != 0;
#define BVG_RL(T, M) do { if (task) { if (gen) hipLaunchKernelGGL((rows_kernel<T, M, true, true>), grid, block, dyn, s, a); \
                                      else hipLaunchKernelGGL((rows_kernel<T, M, false, true>), grid, block, dyn, s, a); } \
                          else { if (gen) hipLaunchKernelGGL((rows_kernel<T, M, true, false>), grid, block, dyn, s, a); \
                                 else hipLaunchKernelGGL((rows_kernel<T, M, false, false>), grid, block, dyn, s, a); } } while (0)
    if (!wide) { if (!materialise) BVG_RL(uint32_t, false); else BVG_RL(uint32_t, true); }
    else { if (!materialise) BVG_RL(uint64_t, false); else BVG_RL(uint64_t, true); }
#undef BVG_RL
}

}  // namespace bvg
