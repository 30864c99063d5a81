// bvg_rows_wg.hip — the row kernel with one WORKGROUP of NW wavefronts per node block (scan mode, default codings).
//
// The single-wavefront row kernel (bvg_rows.hip) is bound by resident waves per CU, and those by LDS: every wave owns a
// list pool.  Here NW wavefronts share one pool, one stream window and one row of <= 64 nodes, so the same LDS feeds NW
// times the lanes:
//   * control is REPLICATED: every wavefront holds the row's nodes one per lane (outdegree, reference, counts, pool
//     slots) and takes every wave-uniform decision (row length, pool allocation, reference levels, task lengths) by
//     itself from identical registers — no messages, no divergence between the wavefronts of a workgroup;
//   * the record headers (reference, copy blocks, intervals: BVG:1015-1058) are parsed by wavefront 0 alone and handed to
//     the others through a small LDS table;
//   * everything that is a flat set of tasks — residual segments (ResidualLongIterator, BVG:902-935), placing the extras,
//     the position tasks of the emission (BVG:1062-1090) — is dealt to all 64*NW lanes.
// Workgroup barriers separate the phases; inside a phase a wavefront only synchronises with itself.
// Anything this kernel does not handle (a list larger than the pool, a record larger than the window, overlapping
// streams, malformed counts) makes the block fail over exactly like the single-wavefront kernel.
#include "bvg_rows_common.h"

namespace bvg {

using namespace rows;

namespace {

enum : uint32_t { WG_FAIL = 1u };

template <typename T, int NW>
__global__ void __launch_bounds__(64 * NW) rows_wg_kernel(DecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];     // pool | scratch | stream window
    __shared__ uint32_t nd_base[kRing];
    __shared__ uint32_t nd_d[kRing];
    __shared__ uint32_t tmaps[NW][64];            // per wavefront: tasks -> (node lane, index inside the node)
    __shared__ uint32_t hd_ref[64], hd_bc[64], hd_ic[64], hd_nres[64], hd_sb[64], hd_ib[64], hd_rel[64], hd_fl[64];   // wavefront 0's header parse
    __shared__ uint32_t wg_k, wg_flags;

    const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    constexpr unsigned NT = 64u * NW;
    const uint32_t bid = a.work_list ? a.work_list[blockIdx.x] : (a.blk_lo + blockIdx.x);
    const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
    if (e <= a.from || s >= a.to || s >= e) return;
    const uint32_t halo = a.blk_halo[bid];
    const uint64_t hmask = a.blk_mask[bid];
    const uint32_t W = (uint32_t)a.window;
    const int64_t hs = s - (int64_t)halo;
    const int64_t rep_lo = s > a.from ? s : a.from, rep_hi = e < a.to ? e : a.to;

    T* const pool = reinterpret_cast<T*>(dyn_lds);
    T* const scr = pool + a.lds_pool_elems;
    const uint32_t* const stage = reinterpret_cast<const uint32_t*>(scr + a.lds_scr_elems);
    uint32_t* const stage_w = const_cast<uint32_t*>(stage);
    uint32_t* const tmap = tmaps[wv];
    const uint32_t CAP = a.lds_pool_elems, SCR = a.lds_scr_elems;
    const uint32_t stage_bits = a.lds_stage_words * 32u;
    const uint32_t zk = (uint32_t)a.cod.zeta_k, minint = (uint32_t)a.min_interval;
    const bool zfast = zk >= 2;
    constexpr uint32_t HS = sizeof(T) * 4;                                    // interval entry: length | position << HS
    const T HM = (T)(((T)1 << HS) - 1);

    for (unsigned i = tid; i < (unsigned)kRing; i += NT) { nd_base[i] = 0; nd_d[i] = 0; }
    if (tid == 0) { wg_flags = 0; wg_k = 0; }
    __syncthreads();

    uint32_t pool_used = 0;
    uint64_t stg_bit0 = 0; uint32_t stg_bits = 0;             // staged window (uniform)
    uint64_t blk_arcs = 0, blk_chk = 0, blk_nodes = 0;        // per wavefront
    unsigned err = 0;
    bool failed = false;
    uint32_t fail_need = 0xFFFFFFFFu;

    const bool sk_have = a.skip_first != nullptr && (!a.skip_fmt || a.skip_fmt[bid] == 1);
    const uint64_t sk_base = sk_have ? a.skip_first[bid] : 0ull;
    const uint32_t sk_n = sk_have ? (uint32_t)(a.skip_first[bid + 1] - sk_base) : 0u;
    uint32_t sk_run = 0;

    int64_t r0 = hs;
    uint64_t off_x = 0, rec_end = 0;
    if (r0 + lane < e) { off_x = a.offsets[r0 + lane]; rec_end = a.offsets[r0 + lane + 1]; }

    while (r0 < e) {
        // ------------------------------------------------------------------ row set-up (replicated)
        const int64_t x = r0 + lane;
        const bool in_range = x < e;
        const uint32_t hbit = x < s ? (uint32_t)(s - 1 - x) : 0;
        const bool needed = in_range && (x >= s || ((hmask >> hbit) & 1ull));
        const uint32_t left = (uint32_t)(e - r0 > 64 ? 64 : e - r0);
        {   // (re)stage the window when this row's records are not covered by it: all lanes of the workgroup copy
            const uint64_t row_lo = lane_get64(off_x, 0);
            const uint64_t row_hi = lane_get64(rec_end, left - 1);
            if (!(row_lo >= stg_bit0 && row_hi + 96 <= stg_bit0 + stg_bits)) {
                __syncthreads();
                const uint64_t b0 = (row_lo >> 3) & ~15ull;
                uint64_t nb = a.padded_bytes > b0 ? a.padded_bytes - b0 : 0;
                if (nb > (stage_bits >> 3)) nb = stage_bits >> 3;
                for (uint32_t c = tid; c < (uint32_t)(nb >> 4); c += NT) {
                    const uint4 v = *reinterpret_cast<const uint4*>(a.graph + b0 + ((uint64_t)c << 4));
                    uint4 w; w.x = __builtin_bswap32(v.x); w.y = __builtin_bswap32(v.y); w.z = __builtin_bswap32(v.z); w.w = __builtin_bswap32(v.w);
                    *reinterpret_cast<uint4*>(&stage_w[c << 2]) = w;
                }
                stg_bit0 = b0 << 3; stg_bits = (uint32_t)(nb << 3);
                __syncthreads();
            }
        }
        const bool inwin = in_range && rec_end + 96 <= stg_bit0 + stg_bits && off_x >= stg_bit0;
        uint32_t kwin;
        {   // contiguous prefix only: rows are cut where the records stop fitting the window
            const uint64_t m = ballot(inwin);
            kwin = m == ~0ull ? 64u : (uint32_t)__ffsll((unsigned long long)~m) - 1u;
            if (kwin > left) kwin = left;
        }
        if (kwin == 0) { failed = true; fail_need = 0xFFFFFFF1u; break; }      // a single record larger than the window
        uint32_t rel = (uint32_t)(off_x - stg_bit0);
        const uint32_t pend = (uint32_t)(rec_end - stg_bit0);
        const uint32_t recrel = rel;
        bool bad = false;
        uint64_t v;
        uint32_t d = 0;
        if (needed && lane < kwin) {                                          // readOutdegree, BVG:654-660
            const uint32_t l = gamma64(win64<LIN>(stage, rel), v);
            bad |= l == 0 || v > 0x7FFFFFFFull; rel += l; d = bad ? 0u : (uint32_t)v;
        }
        const uint32_t dclamp = d > CAP ? CAP + 1 : d;
        const uint32_t incl = wave_incl_scan32(dclamp);
        uint32_t avail = CAP - pool_used;
        const uint32_t total = lane_get(incl, 63);
        if (total > avail && pool_used > 0) {
            // compact: keep only the lists of the last W nodes, moved to the front of the pool (wavefront 0 moves)
            uint32_t my_d = 0, my_base = 0; const int64_t y = r0 - (int64_t)W + (int64_t)lane;
            const bool livelane = lane < W && y >= hs;
            if (livelane) { my_d = nd_d[(uint32_t)y & RM]; my_base = nd_base[(uint32_t)y & RM]; }
            const uint32_t nincl = wave_incl_scan32(my_d);
            const uint32_t nbase = nincl - my_d;
            __syncthreads();
            if (wv == 0) {
                for (uint32_t jn = 0; jn < W && jn < 64; jn++) {
                    const uint32_t src = lane_get(my_base, jn), dst = lane_get(nbase, jn), len = lane_get(my_d, jn);
                    if (src != dst)
                        for (uint32_t t = lane; t < len; t += 64) { const T vv = pool[src + t]; pool[dst + t] = vv; }
                }
                if (livelane) nd_base[(uint32_t)y & RM] = nbase;
            }
            pool_used = lane_get(nincl, 63);
            avail = CAP - pool_used;
            __syncthreads();
        }
        uint32_t k = kwin;
        {   // sized optimistically on the outdegrees (unreferenced lists are not stored), cut exactly after the header parse
            const uint32_t budget = avail + (avail >> 1);
            if (total > budget) { const uint32_t kf = (uint32_t)__popcll(ballot(incl <= budget)); k = kf < k ? kf : k; }
            if (k == 0) k = 1;
        }
        if (wv == 0 && needed && lane < k) nd_d[(uint32_t)x & RM] = d;
        __syncthreads();

        // ------------------------------------------------------------------ phase 1a: headers (wavefront 0)
        uint32_t ref = 0, bc = 0, ic = 0, nres = 0, sb = 0, ib = 0;
        bool malf = false;
        if (wv == 0) {
            int64_t extra = d;
            uint32_t kfail = 0;
            const bool parse0 = needed && lane < k && d > 0;
            if (parse0) {
                if (W > 0) {                                                  // readReference (unary), BVG:692-703
                    const uint64_t w = win64<LIN>(stage, rel);
                    const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u; v = lz; const uint32_t l = lz < 64 ? lz + 1 : 0;
                    bad |= l == 0; rel += l;
                    if (v > W || (int64_t)v > x) { err |= ERR_REF_RANGE; v = 0; }
                    ref = (uint32_t)v;
                }
                if (ref > 0) {                                                // readBlockCount, BVG:728-735
                    const uint32_t l = gamma64(win64<LIN>(stage, rel), v);
                    bad |= l == 0 || v > pend - rel + 1; rel += l; bc = bad ? 0u : (uint32_t)v;
                }
            }
            const uint32_t bincl = wave_incl_scan32(bc > SCR ? SCR + 1 : bc);
            { const uint32_t kb = (uint32_t)__popcll(ballot(bincl <= SCR)); k = kb < k ? kb : k; }
            if (k == 0) { kfail = 0xFFFFFFF3u; k = 1; }                       // one node's copy blocks exceed the scratch area
            sb = bincl - bc;
            uint32_t btot = lane_get(bincl, k - 1);
            if (parse0 && lane < k && !kfail) {
                if (ref > 0) {                                                // copy blocks, BVG:1023-1032
                    int64_t copied = 0, tot = 0;
                    for (uint32_t i = 0; i < bc; i++) {
                        const uint32_t lb = gamma64(win64<LIN>(stage, rel), v);
                        if (lb == 0 || rel > pend) { bad = true; bc = i; break; }
                        rel += lb;
                        const uint32_t b = (uint32_t)v + (i ? 1u : 0u);
                        tot += b;
                        if (!(i & 1)) copied += b;
                        scr[sb + i] = MaskPrefix<T>::pack((uint32_t)tot, (uint32_t)copied);      // prefix form for the position tasks
                    }
                    const int64_t rlen_ = (int64_t)nd_d[(uint32_t)(x - ref) & RM];
                    if (!(bc & 1)) copied += rlen_ - tot;                     // BVG:1030
                    extra = (int64_t)d - copied;
                    if (tot > rlen_) malf = true;
                    if (extra < 0 || copied < 0) { err |= ERR_MALFORMED; extra = 0; malf = true; }
                }
                if (extra > 0 && minint != 0) {                               // interval count: always gamma (BVG:1040)
                    const uint32_t l = gamma64(win64<LIN>(stage, rel), v);
                    bad |= l == 0 || v > (pend - rel) / 2 + 1; rel += l; ic = bad ? 0u : (uint32_t)v;
                }
            }
            const uint32_t iw = lane < k ? 2 * ic : 0u;
            const uint32_t iincl = wave_incl_scan32(iw > SCR ? SCR + 1 : iw);
            for (;;) {                                                        // (see bvg_rows.hip: the blocks may have taken the whole area)
                const uint32_t ki = (uint32_t)__popcll(ballot(btot + iincl <= SCR));
                if (ki != 0 || k <= 1) { k = ki < k ? ki : k; break; }
                k = (k + 1u) >> 1; btot = lane_get(bincl, k - 1);
            }
            if (k == 0) { kfail = 0xFFFFFFF4u; k = 1; }                       // one node's intervals exceed the scratch area
            ib = btot + iincl - iw;
            if (parse0 && lane < k && !kfail) {
                if (ic > 0) {                                                 // intervals, BVG:1042-1058
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
                        extra -= len;
                        if (len > (int64_t)HM) malf = true;                   // does not fit the packed entry (never in an LDS-sized list)
                        scr[ib + 2 * i] = (T)leftv; scr[ib + 2 * i + 1] = (T)len;
                    }
                    if (extra < 0) { err |= ERR_MALFORMED; extra = 0; malf = true; }
                }
                nres = (uint32_t)extra;
            }
            hd_ref[lane] = ref; hd_bc[lane] = bc; hd_ic[lane] = ic; hd_nres[lane] = nres; hd_sb[lane] = sb; hd_ib[lane] = ib; hd_rel[lane] = rel;
            hd_fl[lane] = (uint32_t)bad | ((uint32_t)malf << 1);
            if (lane == 0) { wg_k = k; if (kfail) wg_flags = kfail; }
        }
        __syncthreads();
        if (wv != 0) {
            ref = hd_ref[lane]; bc = hd_bc[lane]; ic = hd_ic[lane]; nres = hd_nres[lane]; sb = hd_sb[lane]; ib = hd_ib[lane]; rel = hd_rel[lane];
            const uint32_t fl = hd_fl[lane]; bad = fl & 1u; malf = (fl >> 1) & 1u;
        }
        k = wg_k;
        {
            const uint32_t fl = wg_flags;
            if (fl) { failed = true; fail_need = fl; break; }
        }
        const bool parse = needed && lane < k && d > 0;

        // ------------------------------------------------------------------ pool allocation (replicated)
        uint64_t refmask = 0;
        for (uint32_t r = 1; r <= W && r < 64; r++) refmask |= ballot(parse && ref == r) >> r;
        uint32_t size = 0, sincl = 0, rtb = 0;
        bool stored = true;
        for (;;) {
            const uint32_t tailstart = k > W ? k - W : 0;
            stored = lane >= tailstart || ((refmask >> lane) & 1ull);
            // lists grow from the bottom of the pool; the row's residual values are parked top-down (+ a guard slot each)
            size = (needed && lane < k && stored) ? dclamp : 0u;
            const uint32_t rsz = (needed && lane < k) ? (nres >= CAP ? CAP + 1 : nres + 1u) : 0u;
            sincl = wave_incl_scan32(size);
            const uint32_t rincl = wave_incl_scan32(rsz);
            rtb = CAP - (rincl > CAP ? CAP : rincl);
            const uint32_t tot = sincl + rincl;
            if (lane_get(tot, k - 1) <= avail) break;
            const uint32_t kf = (uint32_t)__popcll(ballot(tot <= avail && lane < k));
            if (kf == 0) { k = 0; break; }
            k = kf;
        }
        if (k == 0) {                                                         // first node alone overflows the pool
            failed = true;
            uint32_t d0 = lane_get(d, 0); const uint32_t n0 = lane_get(nres, 0);
            if (d0 <= 0x3FFFFFFFu) d0 += (n0 > d0 ? d0 : n0) + 1u;
            fail_need = d0 > 0x3FFFFFFFu ? 0xFFFFFFF2u : d0 + pool_used + (d0 >> 2) + 64;
            break;
        }
        const bool act = needed && lane < k;
        const uint32_t base = pool_used + (sincl - size);
        if (wv == 0 && act) nd_base[(uint32_t)x & RM] = base;
        pool_used += lane_get(sincl, k - 1);
        uint64_t nxt_off = 0, nxt_end = 0;                                    // prefetch the next row's offsets
        {
            const int64_t nx = r0 + k + lane;
            if (nx < e) { nxt_off = a.offsets[nx]; nxt_end = a.offsets[nx + 1]; }
        }
        if (ballot(bad && lane < k)) { failed = true; fail_need = 0xFFFFFFF5u; break; }

        // ------------------------------------------------------------------ phase 1b: residuals, one task per <= 32-gap segment
        bool lbad = false;
        {
            const bool hasres = parse && lane < k && nres > 0;
            const uint32_t cntE = (sk_n != 0 && hasres && nres >= kSkipMin) ? (nres - 1u) / kSkipEvery : 0u;
            uint32_t efirst = 0;
            if (sk_n != 0) {
                const uint32_t cE = (parse && lane < k && nres >= kSkipMin) ? (nres - 1u) / kSkipEvery : 0u;
                const uint32_t eincl = wave_incl_scan32(cE);
                efirst = sk_run + eincl - cE;
                sk_run += lane_get(eincl, 63);
                if (sk_run > sk_n) { failed = true; fail_need = 0xFFFFFFF5u; break; }      // index out of step with the stream
            }
            const uint32_t Tn = hasres ? 1u + cntE : 0u;
            const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, Ttot = lane_get(tincl, 63);
            for (uint32_t pp = 0; pp < Ttot; pp += NT) {
                const uint32_t p0 = pp + wv * 64u;
                {
                    const uint32_t q0 = ts < p0 ? p0 - ts : 0u;
                    const uint32_t q1 = ts >= p0 + 64u ? 0u : (ts + Tn > p0 + 64u ? p0 + 64u - ts : Tn);
                    for (uint32_t q = q0; q < q1; q++) tmap[ts + q - p0] = lane | (q << 8);
                }
                wave_sync();
                const bool tl = p0 + lane < Ttot;
                const uint32_t ent = tl ? tmap[lane] : lane;
                const int nl = (int)(ent & 63u); const uint32_t q = ent >> 8;
                const uint32_t t_rel = __shfl(rel, nl, 64), t_rec = __shfl(recrel, nl, 64), t_pend = __shfl(pend, nl, 64);
                const uint32_t t_nres = __shfl(nres, nl, 64), t_dst = __shfl(rtb, nl, 64), t_ef = __shfl(efirst, nl, 64);
                const uint32_t t_ce = __shfl(cntE, nl, 64);
                if (tl) {
                    const uint32_t t0 = q * kSkipEvery;
                    const uint32_t cnt = q == t_ce ? t_nres - t0 : kSkipEvery;            // the last segment takes the remainder
                    uint32_t trel = t_rel; T r = (T)(r0 + nl);
                    if (q) {
                        const uint64_t en = sk_base + t_ef + q - 1u;
                        trel = t_rec + a.skip_bit[en]; r = reinterpret_cast<const T*>(a.skip_val)[en];
                        if (!(trel > t_rel && trel < t_pend) || trel - t_rec == 0xFFFFu) lbad = true;
                    }
                    T* const tail = pool + t_dst + t0;
                    for (uint32_t i = 0; i < cnt && !lbad; i++) {
                        uint64_t val;
                        const uint32_t len = read_residual<false>(stage, trel, zfast, zk, a.cod.residual, val);
                        if (len == 0) { lbad = true; break; }
                        trel += len;
                        r = (t0 + i) == 0 ? (T)(r + (T)nat2int64(val)) : (T)(r + 1 + (T)val);
                        tail[i] = r;
                        if (trel > t_pend) { err |= ERR_OVERRUN; break; }
                    }
                    if (t0 + cnt == t_nres && trel != t_pend && !lbad) err |= ERR_MALFORMED;
                }
                wave_sync();
            }
            if (wv == 0 && ((act && d == 0) || (parse && nres == 0)) && rel != pend) err |= ERR_MALFORMED;   // SURVEY A.6 self-check
        }
        if (act && wv == 0) pool[rtb + nres] = sentinel<T>();                // guard behind the node's residual positions
        if (ballot(lbad || malf) && lane == 0) wg_flags = 0xFFFFFFF5u;
        __syncthreads();
        {
            const uint32_t fl = wg_flags;
            if (fl) { failed = true; fail_need = fl; break; }
        }

        // ------------------------------------------------------------------ phase 2: emission by output position
        // (see bvg_rows.hip: Z1 places the extras by rank, Z2 runs equal tasks of S output positions)
        const bool rep = act && x >= rep_lo && x < rep_hi;
        uint32_t k0 = 0, k1 = 0;
        if (rep) {
            node_key((uint64_t)x + a.node_base, k0, k1);
        }
        {
            uint32_t rlbN = 0, rlenN = 0;
            if (act && ref > 0) { const int64_t y = x - ref; rlbN = nd_base[(uint32_t)y & RM]; rlenN = nd_d[(uint32_t)y & RM]; }
            const bool own = act && ref == 0;                                 // its parked residuals are its "referenced list"
            if (own) { rlbN = rtb; rlenN = nres; }
            const uint32_t rtbN = own ? rtb + nres : rtb, nresN = own ? 0u : nres;
            const bool inrow = act && ref > 0 && ref <= lane;
            uint32_t lvl = 0;
            for (int it = 0; it < 64; it++) {
                const uint32_t up = __shfl(lvl, inrow ? (int)(lane - ref) : (int)lane, 64);
                const uint32_t nl = inrow ? up + 1 : 0;
                const bool ch = nl != lvl; lvl = nl;
                if (!ballot(ch)) break;
            }
            const bool emitn = act && d > 0;
            bool zbad = false;
            uint64_t remaining = ballot(emitn);
            for (uint32_t L = 0; remaining; L++) {
                const bool mem = emitn && lvl == L;
                remaining &= ~ballot(mem);
                // ---------------- Z1: one lane per extra
                {
                    const uint32_t In = mem ? nresN + ic : 0u;
                    const uint32_t iincl2 = wave_incl_scan32(In), is = iincl2 - In, Itot = lane_get(iincl2, 63);
                    for (uint32_t pp = 0; pp < Itot; pp += NT) {
                        const uint32_t p0 = pp + wv * 64u;
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
                        T vv = 0; uint32_t len = 1, pe = 0; bool isiv = false;
                        if (tl) {
                            uint32_t eb;
                            if (q < t_ic) {                                   // interval q: the intervals and residuals below it
                                isiv = true;
                                vv = scr[t_ib + 2 * q]; len = (uint32_t)(scr[t_ib + 2 * q + 1] & HM);
                                eb = 0;
                                for (uint32_t i = 0; i < q; i++) eb += (uint32_t)(scr[t_ib + 2 * i + 1] & HM);
                                const uint32_t lb = lds_lower_bound<T>(rt, t_nres, vv);
                                if (lb < t_nres && (T)(rt[lb] - vv) < (T)len) zbad = true;         // a residual inside the interval
                                eb += lb;
                            } else {                                          // residual q - ic
                                const uint32_t i = q - t_ic;
                                vv = rt[i]; eb = i;
                                for (uint32_t kk = 0; kk < t_ic; kk++) {
                                    const T lf = scr[t_ib + 2 * kk]; const uint32_t ln = (uint32_t)(scr[t_ib + 2 * kk + 1] & HM);
                                    if (lf <= vv) { eb += ln; if ((T)(vv - lf) < (T)ln) zbad = true; }
                                }
                            }
                            uint32_t t = 0;
                            if (t_rlen) {                                     // copied elements below: rank of the lower bound under the mask
                                const uint32_t qq = lds_lower_bound<T>(rl, t_rlen, vv);
                                uint32_t qn;
                                t = MaskPrefix<T>::rank(scr + t_sb, t_bc, t_rlen, qq, qn);
                                if (qn < t_rlen && (T)(rl[qn] - vv) < (T)len) zbad = true;          // a copied element meets the extra
                            }
                            pe = eb + t;
                            if (pe + len > t_d) { zbad = true; pe = 0; len = 0; }
                        }
                        __syncthreads();                                      // the parked values have been read: positions may replace them
                        if (tl && len) {
                            if (isiv) scr[t_ib + 2 * q + 1] = (T)len | (T)((T)pe << HS);
                            else {
                                if (t_fl & 1u) pool[t_ob + pe] = vv;
                                if (t_fl & 2u) blk_chk += mix_node<T>(t_k1, vv);
                                rt[q - t_ic] = (T)pe;
                            }
                        }
                        wave_sync();
                    }
                }
                __syncthreads();
                // ---------------- Z2: tasks of S output positions over all the lanes of the workgroup
                const uint32_t Wl = wave_sum32(mem ? d : 0u);
                uint32_t S = (Wl + NT - 1u) / NT; if (S < kMinTask) S = kMinTask;
                uint32_t Tn = 0;
                for (int it = 0; it < 6; it++) {
                    Tn = 0;
                    if (mem) { Tn = (uint32_t)((float)d / (float)S); while (Tn * S < d) Tn++; while (Tn > 1u && (Tn - 1u) * S >= d) Tn--; }
                    const uint32_t tt = wave_sum32(Tn);
                    if (tt <= NT || it == 5) break;
                    const uint32_t s2 = (uint32_t)((float)S * (float)tt * (1.0f / (float)NT));
                    S = s2 > S ? s2 : S + 1u;
                }
                const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, Ttot = lane_get(tincl, 63);
                for (uint32_t pp = 0; pp < Ttot; pp += NT) {
                    const uint32_t p0 = pp + wv * 64u;
                    {
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
                    const uint32_t rlast = t_rlen ? t_rlen - 1u : 0u;
                    for (;;) {
                        const bool todo = p < pstop;
                        if (!ballot(todo)) break;
                        if (todo) {
                            if (p == rnext) { ri++; rnext = (uint32_t)rt[ri]; }           // a residual: placed by Z1
                            else {
                                const uint32_t io = p - ivpos;
                                const bool ii = io < ivlen;                               // LongIntervalSequenceIterator.java:71-78
                                const T cv = rl[qcur < rlast ? qcur : rlast];
                                const T ov = ii ? (T)(ivleft + (T)io) : cv;
                                if (t_stored) out[p] = ov;
                                if (t_rep) blk_chk += mix_node<T>(t_k1, ov);
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
                }
                if (ballot(zbad) && lane == 0) wg_flags = 0xFFFFFFF5u;
                __syncthreads();                                              // the level's lists are complete
            }
            {
                const uint32_t fl = wg_flags;
                if (fl) { failed = true; fail_need = fl; break; }
            }
            if (rep && wv == 0) { blk_arcs += d; blk_nodes += 1; blk_chk += mix_node_const(k0, k1, a.node_base, d); }
        }

        __syncthreads();
        r0 += k;
        off_x = nxt_off; rec_end = nxt_end;
    }

    if (failed) {
        if (tid == 0) {
            uint32_t slot = atomicAdd(a.fail_count, 1u);
            if (slot < a.fail_cap) { a.fail_list[slot] = bid; if (a.fail_need) a.fail_need[slot] = fail_need; }
        }
        return;
    }
    err = wave_or32(err);
    blk_arcs = wave_sum64(blk_arcs); blk_chk = wave_sum64(blk_chk); blk_nodes = wave_sum64(blk_nodes);
    if (lane == 0) {
        unsigned long long* const accs = a.acc + (size_t)(bid & a.acc_mask) * kAccStride;   // this block's result stripe
        if (blk_arcs) atomicAdd(&accs[0], (unsigned long long)blk_arcs);
        if (blk_chk) atomicAdd(&accs[1], (unsigned long long)blk_chk);
        if (blk_nodes) atomicAdd(&accs[2], (unsigned long long)blk_nodes);
        if (err) atomicOr(&accs[3], (unsigned long long)err);
    }
}

}  // namespace

// LDS the kernel declares statically, per workgroup (the host sizes the dynamic part around it)
size_t rows_wg_static_lds(int nw) { return (size_t)(2 * kRing + nw * 64 + 8 * 64 + 2) * 4; }

void launch_rows_wg_decode(const DecodeArgs& a, uint32_t nblocks, int nw, hipStream_t s) {
    if (nblocks == 0) return;
    dim3 grid(nblocks);
    const size_t dyn = (size_t)(a.lds_pool_elems + a.lds_scr_elems) * 4 + (size_t)a.lds_stage_words * 4;
    if (nw == 2) hipLaunchKernelGGL((rows_wg_kernel<uint32_t, 2>), grid, dim3(128), dyn, s, a);
    else if (nw == 8) hipLaunchKernelGGL((rows_wg_kernel<uint32_t, 8>), grid, dim3(512), dyn, s, a);
    else hipLaunchKernelGGL((rows_wg_kernel<uint32_t, 4>), grid, dim3(256), dyn, s, a);
}

}  // namespace bvg
