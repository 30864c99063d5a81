// bvg_flow.hip — the "flow" scan kernel: an EXPERIMENTAL tier 0 for full scans of graphs with BVGraph's default codings
// (opt-in: BVG_FLOW=1; parity-green, 0.35-0.5x the row kernel's rate today -- profiles/r02).
//
// Why it exists (profiles/r02): the row kernel is latency bound per wavefront and its throughput is LINEAR in the wavefronts
// resident on a CU (1..10 measured), and those are bound by LDS: a row of 64 lists, the parked residuals of the row, the copy
// blocks and the stream window make ~16-20 KiB per wavefront.  This kernel keeps in LDS only what must be randomly accessed
// later -- the successor lists of the last W nodes that some later node really copies from (BVG:1062-1090) -- and moves everything
// that is produced once and consumed once in order (parsed headers, copy blocks, intervals, decoded residual values) through a
// per-wavefront scratch area in global memory, which stays in L2 / Infinity Cache between its write and its read.
//
// One wavefront per node block (persistent: a wavefront takes block after block), two stages per block:
//   stage 1  (everything that does not depend on other nodes' lists) rows of up to 64 nodes, one node per lane, exactly the
//            parse of the row kernel: outdegree gamma, reference unary, copy blocks gamma (stored in prefix form), intervals gamma,
//            residual gaps zeta_k cut at the skip index into tasks for all lanes (BVG:1003-1064, 902-935).  Residual values and
//            interval elements are summed into the checksum right here; values / blocks / intervals go to the scratch area.
//   stage 2  (reference resolution) node after node, ALL 64 lanes on one node: the kept elements of the referenced list
//            (MaskedLongIterator.java:73-100) are summed flat, 64 list positions per step; every residual and interval is located in
//            the referenced list by binary search, which detects the equal heads that MergedLongIterator.java:85-89 would emit once
//            (such blocks go to the row kernel / literal tier) and, only for a list that a later node copies from, gives the output
//            positions with which the merged list is written to the LDS ring.
// Lists too long for the ring live in the scratch area instead (flat reads are coalesced either way).  Whatever does not fit
// the fixed capacities (more than kFlowNodes nodes in a block, lists over kFlowMaxList, windows over 64, ...) fails over to the row
// kernel through the usual fail list.
#include "bvg_rows_common.h"

#include <type_traits>

namespace bvg {

namespace {

using namespace rows;

typedef uint32_t T;                                     // 32-bit successors (graphs of < 2^31 nodes)
constexpr uint32_t kFlowNodes = 2048;                   // nodes (halo included) per block
constexpr uint32_t kFlowBlk = 16384;                    // copy-block + interval words per block
constexpr uint32_t kFlowRes = 32768;                    // residual values per block
constexpr uint32_t kFlowMaxList = 8192;                 // longest list held (scratch-backed)
constexpr uint32_t kFlowAux = 1024;                     // LDS words: stream window (stage 1) | copy blocks, intervals, insertion ranks (stage 2)
constexpr uint32_t kAuxBlk = 256, kAuxIv = 128, kAuxC = kFlowAux - 2 * (kAuxBlk + kAuxIv);   // (two payload buffers)

struct Hdr { uint32_t d, pk, nres, boff, ioff, roff, flags, pad; };   // pk = ref | ic << 8 | bc << 16; flags: 1 needed, 2 reported, 4 some later node copies from it
static_assert(sizeof(Hdr) == 32, "Hdr");

__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace

// scratch-backed lists: a circular area of (window + 2) longest lists, so an allocation never wraps onto a list still in the window
static uint32_t flow_glist_elems(int window) { return (uint32_t)(window + 2) * kFlowMaxList; }
size_t flow_scratch_bytes_per_wave(int window) {
    return (size_t)kFlowNodes * sizeof(Hdr) + ((size_t)kFlowBlk + kFlowRes + flow_glist_elems(window)) * sizeof(uint32_t);
}

namespace {

__global__ void __launch_bounds__(64, 5) flow_kernel(DecodeArgs a, uint32_t nwork, uint8_t* scratch, uint64_t scratch_stride, uint32_t ring_cap, uint32_t gl_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];     // list ring | aux
    __shared__ uint32_t nd_base[kRing];
    __shared__ uint32_t nd_d[kRing];               // bit 31: the list lives in the scratch area
    __shared__ uint32_t rtmap[128];

    const unsigned lane = threadIdx.x;
    T* const ring = reinterpret_cast<T*>(dyn_lds);
    uint32_t* const aux = reinterpret_cast<uint32_t*>(ring + ring_cap);
    uint8_t* const my = scratch + (size_t)blockIdx.x * scratch_stride;
    Hdr* const hdr = reinterpret_cast<Hdr*>(my);
    uint32_t* const gblk = reinterpret_cast<uint32_t*>(hdr + kFlowNodes);
    uint32_t* const gres = gblk + kFlowBlk;
    uint32_t* const glist = gres + kFlowRes;
    const uint32_t W = (uint32_t)a.window, zk = (uint32_t)a.cod.zeta_k, minint = (uint32_t)a.min_interval;
    const bool zfast = zk >= 2;
    const uint32_t stage_bits = kFlowAux * 32u;

    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const uint32_t bid = a.work_list ? a.work_list[wi] : (a.blk_lo + wi);
        const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
        if (e <= a.from || s >= a.to || s >= e) continue;
        const uint32_t halo = a.blk_halo[bid];
        const uint64_t hmask = a.blk_mask[bid];
        const int64_t hs = s - (int64_t)halo;
        const int64_t rep_lo = s > a.from ? s : a.from, rep_hi = e < a.to ? e : a.to;
        uint64_t blk_arcs = 0, blk_chk = 0, blk_nodes = 0;
        unsigned err = 0;
        bool failed = false;
        uint32_t fail_need = 0xFFFFFFF5u;
        wave_sync();
        for (unsigned i = lane; i < (unsigned)kRing; i += 64) { nd_base[i] = 0; nd_d[i] = 0; }
        if ((uint64_t)(e - hs) > kFlowNodes) { failed = true; fail_need = 0xFFFFFFF6u; }

        // residual skip index of this block
        const bool sk_have = a.skip_first != nullptr && (!a.skip_fmt || a.skip_fmt[bid] == 1);
        const uint64_t sk_base = sk_have ? a.skip_first[bid] : 0ull;
        const uint32_t sk_n = sk_have ? (uint32_t)(a.skip_first[bid + 1] - sk_base) : 0u;
        uint32_t sk_run = 0;

        // =============================================================================== stage 1: parse every record of the block
        uint32_t boff_run = 0, roff_run = 0;
        uint64_t stg_bit0 = 0; uint32_t stg_bits = 0;
        int64_t r0 = hs;
        uint64_t off_x = 0, rec_end = 0;
        if (!failed && r0 + lane < e) { off_x = a.offsets[r0 + lane]; rec_end = a.offsets[r0 + lane + 1]; }
        while (!failed && r0 < e) {
            const int64_t x = r0 + lane;
            const bool in_range = x < e;
            const uint32_t hbit = x < s ? (uint32_t)(s - 1 - x) : 0;
            const bool needed = in_range && (x >= s || ((hmask >> hbit) & 1ull));
            const uint32_t left = (uint32_t)(e - r0 > 64 ? 64 : e - r0);
            {
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
                        *reinterpret_cast<uint4*>(&aux[c << 2]) = w;
                    }
                    stg_bit0 = b0 << 3; stg_bits = (uint32_t)(nb << 3);
                    wave_sync();
                }
            }
            const uint32_t* const stage = aux;
            const bool inwin = in_range && rec_end + 96 <= stg_bit0 + stg_bits && off_x >= stg_bit0;
            uint32_t k;
            {
                const uint64_t m = ballot(inwin);
                k = m == ~0ull ? 64u : (uint32_t)__ffsll((unsigned long long)~m) - 1u;
                if (k > left) k = left;
            }
            if (k == 0) { failed = true; fail_need = 0xFFFFFFF1u; break; }
            uint32_t rel = (uint32_t)(off_x - stg_bit0);
            const uint32_t pend = (uint32_t)(rec_end - stg_bit0);
            const uint32_t recrel = rel;
            bool bad = false;
            uint64_t v;
            uint32_t d = 0;
            const bool mine = needed && lane < k;
            if (mine) {                                                          // readOutdegree, BVG:654-660
                const uint32_t l = gamma64(win64<LIN>(stage, rel), v);
                bad |= l == 0 || v > 0x7FFFFFFFull; rel += l; d = bad ? 0u : (uint32_t)v;
                if (d > kFlowMaxList) bad = true;
            }
            if (mine) nd_d[(uint32_t)x & RM] = d;
            wave_sync();
            uint32_t ref = 0, bc = 0, ic = 0, nres = 0, ivtot = 0;
            int64_t extra = d;
            const bool parse = mine && d > 0 && !bad;
            if (parse) {
                if (W > 0) {                                                      // readReference, BVG:692-703
                    const uint64_t w = win64<LIN>(stage, rel);
                    const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u; v = lz;
                    const uint32_t l = lz < 64 ? lz + 1 : 0;
                    bad |= l == 0; rel += l;
                    if (v > W || (int64_t)v > x) { err |= ERR_REF_RANGE; v = 0; }
                    ref = (uint32_t)v;
                }
                if (ref > 0) {                                                    // readBlockCount, BVG:728-735
                    const uint32_t l = gamma64(win64<LIN>(stage, rel), v);
                    bad |= l == 0 || v > pend - rel + 1 || v > kAuxBlk; rel += l; bc = bad ? 0u : (uint32_t)v;
                }
            }
            // copy blocks to the scratch area, already in prefix form (end position | kept so far << 16: MaskPrefix)
            // (stage 1 does not use the list ring: it serves as the staging area from which blocks, intervals, residual values and
            //  headers leave for the scratch area in coalesced 256-byte stores -- one lane per node writing its own run would be 64
            //  separate partial-line stores per instruction)
            const uint32_t bincl = wave_incl_scan32(bc);
            const uint32_t sbl = bincl - bc, sb = boff_run + sbl;
            if (boff_run + lane_get(bincl, 63) > kFlowBlk || lane_get(bincl, 63) > ring_cap) { failed = true; fail_need = 0xFFFFFFF3u; break; }
            bool malf = false;
            if (parse && !bad) {
                if (ref > 0) {
                    int64_t copied = 0, tot = 0;
                    for (uint32_t i = 0; i < bc; i++) {
                        const uint32_t lb = gamma64(win64<LIN>(stage, rel), v);
                        if (lb == 0 || rel > pend) { bad = true; break; }
                        rel += lb;
                        const uint32_t b = (uint32_t)v + (i ? 1u : 0u);
                        tot += b; if (!(i & 1)) copied += b;
                        if (tot > 0xFFFF) { bad = true; break; }
                        ring[sbl + i] = (uint32_t)tot | ((uint32_t)copied << 16);
                    }
                    const int64_t rlen_ = (int64_t)(nd_d[(uint32_t)(x - ref) & RM] & 0x7FFFFFFFu);
                    if (!(bc & 1)) copied += rlen_ - tot;                         // BVG:1030
                    extra = (int64_t)d - copied;
                    if (tot > rlen_ || extra < 0 || copied < 0) malf = true;      // streams the position logic cannot take: the literal tier decides
                }
                if (!malf && extra > 0 && minint != 0) {                          // always gamma (BVG:1040)
                    const uint32_t l = gamma64(win64<LIN>(stage, rel), v);
                    bad |= l == 0 || v > (pend - rel) / 2 + 1 || v > kAuxIv / 2; rel += l; ic = bad ? 0u : (uint32_t)v;
                }
            }
            const uint32_t btot = lane_get(bincl, 63);
            const uint32_t iincl = wave_incl_scan32(2 * ic);
            const uint32_t ibl = btot + iincl - 2 * ic, ib = boff_run + ibl;
            if (boff_run + btot + lane_get(iincl, 63) > kFlowBlk || btot + lane_get(iincl, 63) > ring_cap) { failed = true; fail_need = 0xFFFFFFF3u; break; }
            const bool rep = mine && x >= rep_lo && x < rep_hi;
            uint32_t k0 = 0, k1 = 0;
            if (rep) {
                node_key((uint64_t)x + a.node_base, k0, k1);
            }
            if (parse && !bad && !malf && ic > 0) {                               // intervals (BVG:1042-1058): they fix the number of residuals
                int64_t prev = 0;
                for (uint32_t i = 0; i < ic; i++) {
                    uint64_t v1, v2;
                    const uint32_t l1 = gamma64(win64<LIN>(stage, rel), v1);
                    const uint32_t l2 = gamma64(win64<LIN>(stage, rel + l1), v2);
                    if (l1 == 0 || l2 == 0 || rel > pend) { bad = true; break; }
                    rel += l1 + l2;
                    const int64_t leftv = i == 0 ? x + nat2int64(v1) : prev + 1 + (int64_t)v1;
                    const int64_t len = (int64_t)v2 + minint;
                    if (len > 0xFFFF || leftv < 0 || leftv + len > 0x7FFFFFFFll) { malf = true; break; }
                    prev = leftv + len; extra -= len; ivtot += (uint32_t)len;
                    ring[ibl + 2 * i] = (uint32_t)leftv; ring[ibl + 2 * i + 1] = (uint32_t)len;
                }
                if (extra < 0) malf = true;
            }
            if (parse && !bad && !malf) nres = (uint32_t)extra;
            {   // copy blocks and intervals of the row: LDS -> scratch area, coalesced
                const uint32_t tot = btot + lane_get(iincl, 63);
                wave_sync();
                for (uint32_t t = lane; t < tot; t += 64) gblk[boff_run + t] = ring[t];
                wave_sync();
            }
            // residual values: offsets in the scratch area
            const uint32_t rincl = wave_incl_scan32(nres);
            const uint32_t rb = roff_run + rincl - nres;
            if (roff_run + lane_get(rincl, 63) > kFlowRes) { failed = true; fail_need = 0xFFFFFFF4u; break; }
            if (ballot(malf)) { failed = true; fail_need = 0xFFFFFFF5u; break; }
            // which lists does a later node of this block copy from?  (inside the row: ballots; earlier rows: a flag in their header)
            uint64_t refmask = 0;
            for (uint32_t r = 1; r <= W && r < 64; r++) refmask |= ballot(parse && ref == r) >> r;
            if (parse && ref > lane) atomicOr(&hdr[(uint32_t)(x - ref - hs)].flags, 4u);   // (a header of an earlier row)
            // prefetch the next row's offsets
            uint64_t nxt_off = 0, nxt_end = 0;
            { const int64_t nx = r0 + k + lane; if (nx < e) { nxt_off = a.offsets[nx]; nxt_end = a.offsets[nx + 1]; } }
            // ---- residuals (ResidualLongIterator, BVG:902-935): values staged in LDS chunk by chunk, summed on the way
            uint32_t cntE = (parse && !bad && sk_n != 0 && nres >= kSkipMin) ? (nres - 1u) / kSkipEvery : 0u;
            const uint32_t eincl = wave_incl_scan32(cntE);
            const uint32_t efirst = sk_run + eincl - cntE;
            sk_run += lane_get(eincl, 63);
            if (sk_n != 0 && sk_run > sk_n) { failed = true; break; }
            uint64_t chk = 0;
            bool rfail = false;
            for (uint32_t c0 = 0; c0 < k;) {
                // lanes [c0, c1): as many nodes as the staging area holds residuals for
                const uint32_t cb = lane_get(rincl - nres, c0);
                const bool fits = lane >= c0 && lane < k && rincl - cb <= ring_cap;
                const uint32_t c1 = c0 + (uint32_t)__popcll(ballot(fits));
                if (c1 == c0) { rfail = true; break; }                           // one node's residuals exceed the staging area
                const bool inck = lane >= c0 && lane < c1;
                const uint32_t rbl = rincl - nres - cb;                          // where this node's values go in the staging area
                const uint32_t ctot = lane_get(rincl, c1 - 1) - cb;
                const bool cparse = parse && !bad && inck;
                if (sk_n != 0 && ballot(inck && cntE != 0)) {
                    const bool hasres = cparse && nres > 0;
                    const uint32_t Tn = hasres ? cntE + 1u : 0u;
                    const uint32_t tincl = wave_incl_scan32(Tn), ts = tincl - Tn, Ttot = lane_get(tincl, 63);
                    bool tbad = false;
                    for (uint32_t p0 = 0; p0 < Ttot; p0 += 64) {
                        {
                            const uint32_t q0 = ts < p0 ? p0 - ts : 0u;
                            const uint32_t q1 = ts >= p0 + 64u ? 0u : (ts + Tn > p0 + 64u ? p0 + 64u - ts : Tn);
                            for (uint32_t q = q0; q < q1; q++) rtmap[ts + q - p0] = lane | (q << 8);
                        }
                        wave_sync();
                        const bool tl = p0 + lane < Ttot;
                        const uint32_t ent = tl ? rtmap[lane] : lane;
                        const int nl = (int)(ent & 63u); const uint32_t q = ent >> 8;
                        const uint32_t t_rel = __shfl(rel, nl, 64), t_rec = __shfl(recrel, nl, 64), t_pend = __shfl(pend, nl, 64);
                        const uint32_t t_nres = __shfl(nres, nl, 64), t_dst = __shfl(rbl, nl, 64), t_ef = __shfl(efirst, nl, 64);
                        const uint32_t t_k1 = __shfl(k1, nl, 64);
                        const uint32_t t0 = q * kSkipEvery;
                        const uint32_t t_ce = t_nres >= kSkipMin ? (t_nres - 1u) / kSkipEvery : 0u;
                        const uint32_t cnt0 = tl ? (q == t_ce ? t_nres - t0 : kSkipEvery) : 0u;
                        uint32_t cnt = cnt0;
                        uint32_t trel = tl ? t_rel : 0u; T r = (T)(r0 + nl);
                        if (tl && q) {
                            const uint64_t en = sk_base + t_ef + q - 1u;
                            trel = t_rec + a.skip_bit[en]; r = reinterpret_cast<const T*>(a.skip_val)[en];
                            if (!(trel > t_rel && trel < t_pend) || trel - t_rec == 0xFFFFu) { tbad = true; cnt = 0; trel = 0; }
                        }
                        for (uint32_t i = 0;; i++) {
                            const bool on = i < cnt;
                            if (!ballot(on)) break;
                            uint32_t len = 0; uint64_t val = 0;
                            if (zfast) { uint32_t v32; len = zeta_fast32(win32<LIN>(stage, trel), zk, v32); val = v32; }
                            if (ballot(on && len == 0)) {
                                if (on && len == 0) { len = zeta64(win64<LIN>(stage, trel), zk, val); if (len == 0) { tbad = true; cnt = 0; } }
                            }
                            if (on && len) {
                                trel += len;
                                r = (t0 + i) == 0 ? (T)(r + (T)nat2int64(val)) : (T)(r + 1 + (T)val);
                                ring[t_dst + t0 + i] = r;
                                chk += mix_node<T>(t_k1, r);      // (k1 = 0 for nodes outside the reported range)
                                if (trel > t_pend) { err |= ERR_OVERRUN; cnt = 0; }
                            }
                        }
                        if (tl && t0 + cnt0 == t_nres && cnt && trel != t_pend && !tbad) err |= ERR_MALFORMED;
                        wave_sync();
                    }
                    if (inck && ((mine && d == 0) || (parse && nres == 0)) && rel != pend && !bad) err |= ERR_MALFORMED;
                    bad |= tbad;
                } else if (cparse) {
                    T r = (T)x;
                    for (uint32_t t = 0; t < nres; t++) {
                        uint64_t val;
                        const uint32_t len = read_residual<false>(stage, rel, zfast, zk, a.cod.residual, val);
                        if (len == 0) { bad = true; break; }
                        rel += len;
                        r = t == 0 ? (T)(r + (T)nat2int64(val)) : (T)(r + 1 + (T)val);
                        ring[rbl + t] = r;
                        chk += mix_node<T>(k1, r);
                        if (rel > pend) { err |= ERR_OVERRUN; break; }
                    }
                    if (rel != pend && !bad) err |= ERR_MALFORMED;               // SURVEY A.6 self-check
                } else if (inck && mine && d == 0 && rel != pend && !bad) err |= ERR_MALFORMED;
                wave_sync();
                for (uint32_t t = lane; t < ctot; t += 64) gres[roff_run + cb + t] = ring[t];     // LDS -> scratch area, coalesced
                wave_sync();
                c0 = c1;
            }
            if (rfail) { failed = true; fail_need = 0xFFFFFFF8u; break; }
            if (ballot(bad && lane < k)) { failed = true; fail_need = d > kFlowMaxList ? d + (d >> 2) + 64u : 0xFFFFFFF5u; break; }
            blk_chk += chk;
            {   // headers of the row: LDS -> scratch area, coalesced (8 words per node, nodes are consecutive)
                if (lane < k && in_range) {
                    uint32_t* const hl = ring + lane * 8u;
                    hl[0] = d; hl[1] = ref | (ic << 8) | (bc << 16); hl[2] = nres; hl[3] = sb; hl[4] = ib; hl[5] = rb;
                    hl[6] = (mine ? 1u : 0u) | (rep ? 2u : 0u) | (((refmask >> lane) & 1ull) ? 4u : 0u); hl[7] = 0;
                }
                wave_sync();
                uint32_t* const hg = reinterpret_cast<uint32_t*>(hdr + (uint32_t)(r0 - hs));
                for (uint32_t t = lane; t < k * 8u; t += 64) hg[t] = ring[t];
                wave_sync();
            }
            if (rep) { blk_arcs += d; blk_nodes += 1; blk_chk += mix_node_const(k0, k1, a.node_base, d); }   // (round 5's checksum: k1 * successor per arc + d * (k1 * base + k0) per node)
            boff_run += btot + lane_get(iincl, 63);
            roff_run += lane_get(rincl, 63);
            wave_sync();
            r0 += k;
            off_x = nxt_off; rec_end = nxt_end;
        }

        // =============================================================================== stage 2: reference resolution, node after node
        if (!failed) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                 // the scratch area is read back with L1-bypassing loads
            wave_sync();
            uint32_t ring_used = 0, gl_used = 0;
            // aux in stage 2: two buffers of {copy blocks, intervals} (the next node's are fetched while this one is worked on) + insertion ranks
            uint32_t* const acr = aux + 2 * (kAuxBlk + kAuxIv);
            const uint32_t nn = (uint32_t)(e - hs);
            uint64_t chk = 0;
            bool zbad = false;
            uint32_t buf = 0;
            for (uint32_t i0 = 0; i0 < nn && !failed; i0 += 64) {
                // the headers of 64 nodes: one per lane, then handed round through SGPRs (no dependent global read per node)
                uint32_t H[7];
                {
                    const bool hl = i0 + lane < nn;
                    const uint32_t* hp = reinterpret_cast<const uint32_t*>(hdr + (hl ? i0 + lane : i0));
#pragma unroll
                    for (int w = 0; w < 7; w++) H[w] = ld_sc1(hp + w);
                    if (!hl) { H[0] = 0; H[6] = 0; }
                }
                // nodes with something left to do: a reference, intervals, or a list that a later node copies from
                const bool actl = (H[6] & 1u) && H[0] != 0 && ((H[1] & 0xFFu) != 0 || ((H[1] >> 8) & 0xFFu) != 0 || (H[6] & 4u));
                if (i0 + lane < nn) nd_base[(uint32_t)(hs + i0 + lane) & RM] = 0xFFFFFFFFu;   // (stage 1 left its own values in the ring tables)
                uint64_t am = ballot(actl);
                // payload registers of the node that comes next (prefetched), valid when pre_j == j
                uint32_t pb[4] = {0, 0, 0, 0}, pi[2] = {0, 0}; int pre_j = -1;
                auto fetch = [&](uint32_t j) {                                   // issue the loads of node j's copy blocks and intervals
                    const uint32_t pk = lane_get(H[1], j), bo = lane_get(H[3], j), io = lane_get(H[4], j);
                    const uint32_t bcj = pk >> 16, icj = (pk >> 8) & 0xFFu;
#pragma unroll
                    for (int w = 0; w < 4; w++) pb[w] = (lane + 64u * w < bcj) ? ld_sc1(gblk + bo + lane + 64u * w) : 0u;
#pragma unroll
                    for (int w = 0; w < 2; w++) pi[w] = (lane + 64u * w < 2u * icj) ? ld_sc1(gblk + io + lane + 64u * w) : 0u;
                    pre_j = (int)j;
                };
                wave_sync();
                while (am && !failed) {
                    const uint32_t j = (uint32_t)__ffsll((unsigned long long)am) - 1u;
                    am &= am - 1;
                    const uint32_t i = i0 + j;
                    const uint32_t d = lane_get(H[0], j), pk = lane_get(H[1], j), nres = lane_get(H[2], j), roff = lane_get(H[5], j), fl = lane_get(H[6], j);
                    const uint32_t ref = pk & 0xFFu, ic = (pk >> 8) & 0xFFu, bc = pk >> 16;
                    if (pre_j != (int)j) fetch(j);
                    // park the payload in LDS, then start the next node's loads: they fly while this node is worked on
                    uint32_t* const ablk = aux + buf * (kAuxBlk + kAuxIv); uint32_t* const aiv = ablk + kAuxBlk; buf ^= 1u;
#pragma unroll
                    for (int w = 0; w < 4; w++) if (lane + 64u * w < bc) ablk[lane + 64u * w] = pb[w];
#pragma unroll
                    for (int w = 0; w < 2; w++) if (lane + 64u * w < 2u * ic) aiv[lane + 64u * w] = pi[w];
                    // the first 64 residuals of this node (more are read in the loop)
                    const uint32_t r_first = (lane < nres) ? ld_sc1(gres + roff + lane) : 0u;
                    if (am) fetch((uint32_t)__ffsll((unsigned long long)am) - 1u);
                    wave_sync();
                    const int64_t x = hs + i;
                    const bool rep = fl & 2u, stored = fl & 4u;
                    uint32_t k0 = 0, k1 = 0;
                    if (rep) {
                        node_key((uint64_t)x + a.node_base, k0, k1);
                    }
                    uint32_t ivtot = 0;
                    for (uint32_t kk = 0; kk < ic; kk++) {                        // interval elements into the checksum
                        const uint32_t lf = aiv[2 * kk], ln = aiv[2 * kk + 1];
                        if (rep) for (uint32_t jj = lane; jj < ln; jj += 64) chk += mix_node<T>(k1, (T)(lf + jj));
                        ivtot += ln;
                    }
                    const uint32_t kept = d - nres - ivtot;                       // copied elements (0 without a reference)
                    // where the merged list goes, if a later node copies from it
                    uint32_t ob = 0; bool ogl = false;
                    if (stored) {
                        if (d <= (ring_cap >> 2)) {
                            if (ring_used + d > ring_cap) {                       // compact the ring: keep the lists of the last W nodes
                                uint32_t my_d = 0, my_base = 0xFFFFFFFFu; const int64_t y = x - (int64_t)W + (int64_t)lane;
                                const bool live = lane < W && y >= hs;
                                if (live) { const uint32_t dd = nd_d[(uint32_t)y & RM]; my_base = nd_base[(uint32_t)y & RM]; if (!(dd & 0x80000000u)) my_d = dd; }
                                const bool in_ring = live && my_d != 0 && my_base != 0xFFFFFFFFu;
                                const uint32_t sz = in_ring ? my_d : 0u;
                                const uint32_t nincl = wave_incl_scan32(sz), nbase = nincl - sz;
                                for (uint32_t jn = 0; jn < W && jn < 64; jn++) {
                                    const uint32_t src = lane_get(my_base, jn), dst = lane_get(nbase, jn), len = lane_get(sz, jn);
                                    if (len && src != dst)
                                        for (uint32_t t0 = 0; t0 < len; t0 += 64) { const uint32_t t = t0 + lane; T vv = 0; if (t < len) vv = ring[src + t]; wave_sync(); if (t < len) ring[dst + t] = vv; }
                                }
                                if (in_ring) nd_base[(uint32_t)y & RM] = nbase;
                                ring_used = lane_get(nincl, 63);
                                wave_sync();
                            }
                            if (ring_used + d > ring_cap) { failed = true; fail_need = 0xFFFFFFF7u; break; }
                            ob = ring_used; ring_used += d;
                        } else {                                                  // a long list: scratch-backed, circular
                            if (gl_used + d > gl_cap) gl_used = 0;
                            ob = gl_used; gl_used += d; ogl = true;
                        }
                    }
                    // the referenced list
                    uint32_t rlen = 0, rbs = 0; bool rgl = false;
                    if (ref) {
                        const uint32_t yy = (uint32_t)(x - ref) & RM;
                        const uint32_t dd = nd_d[yy];
                        rlen = dd & 0x7FFFFFFFu; rgl = dd & 0x80000000u; rbs = nd_base[yy];
                        if (rbs == 0xFFFFFFFFu) { failed = true; break; }        // not materialised (cannot happen: the flag says it is copied from)
                    }
                    auto rd = [&](uint32_t q) -> T { return rgl ? (T)ld_sc1(glist + rbs + q) : ring[rbs + q]; };
                    auto wr = [&](uint32_t p, T v) { if (ogl) glist[ob + p] = v; else ring[ob + p] = v; };
                    auto resv = [&](uint32_t jj) -> T { return jj < 64u ? (T)__shfl(r_first, (int)jj, 64) : (T)ld_sc1(gres + roff + jj); };   // (uniform jj only)
                    if (ref == 0) {
                        // no reference: the list is the residuals merged with the intervals (BVG:1087-1089)
                        if (ic) {
                            for (uint32_t j0 = 0; j0 < nres; j0 += 64) {
                                const uint32_t jj = j0 + lane;
                                if (jj < nres) {
                                    const T r = j0 == 0 ? (T)r_first : (T)ld_sc1(gres + roff + jj);
                                    uint32_t below = 0;
                                    for (uint32_t kk = 0; kk < ic; kk++) { const uint32_t lf = aiv[2 * kk], ln = aiv[2 * kk + 1]; if (lf <= r) { below += ln; if (r - lf < ln) zbad = true; } }
                                    if (stored) wr(jj + below, r);
                                }
                            }
                            if (stored) {                                         // interval elements: behind the residuals below the interval's left end
                                uint32_t before = 0;
                                for (uint32_t kk = 0; kk < ic; kk++) {
                                    const uint32_t lf = aiv[2 * kk], ln = aiv[2 * kk + 1];
                                    uint32_t lo = 0, hi = nres;                   // residuals < lf: the same search in every lane
                                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (resv(mid) < (T)lf) lo = mid + 1; else hi = mid; }
                                    for (uint32_t jj = lane; jj < ln; jj += 64) wr(lo + before + jj, (T)(lf + jj));
                                    before += ln;
                                }
                            }
                        } else if (stored) {
                            for (uint32_t j0 = 0; j0 < nres; j0 += 64) { const uint32_t jj = j0 + lane; if (jj < nres) wr(jj, j0 == 0 ? (T)r_first : (T)ld_sc1(gres + roff + jj)); }
                        }
                    } else {
                        const bool tailkeep = !(bc & 1u);
                        if (stored && nres > kAuxC) { failed = true; fail_need = 0xFFFFFFF8u; break; }
                        // ---- every residual: where does it fall in the referenced list?  (equal heads -> literal tier; insertion ranks kept for stored lists)
                        for (uint32_t j0 = 0; j0 < nres; j0 += 64) {
                            const uint32_t jj = j0 + lane; const bool on = jj < nres;
                            const T r = on ? (j0 == 0 ? (T)r_first : (T)ld_sc1(gres + roff + jj)) : (T)0;
                            uint32_t lo = 0, hi = on ? rlen : 0u;                 // lower bound of r in the referenced list
                            while (ballot(lo < hi)) { if (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rd(mid) < r) lo = mid + 1; else hi = mid; } }
                            if (on) {
                                uint32_t qn; const uint32_t c = MaskPrefix<T>::rank((const T*)ablk, bc, rlen, lo, qn);
                                if (qn < rlen && rd(qn) == r) zbad = true;        // a copied element equals the residual
                                uint32_t below = 0;
                                for (uint32_t kk = 0; kk < ic; kk++) { const uint32_t lf = aiv[2 * kk], ln = aiv[2 * kk + 1]; if (lf <= r) { below += ln; if (r - lf < ln) zbad = true; } }
                                if (stored) { acr[jj] = c; wr(jj + below + c, r); }
                            }
                        }
                        // intervals: kept elements below their left end; equal heads with copied elements
                        uint32_t before = 0;
                        for (uint32_t kk = 0; kk < ic; kk++) {
                            const uint32_t lf = aiv[2 * kk], ln = aiv[2 * kk + 1];
                            uint32_t lo = 0, hi = rlen;
                            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rd(mid) < (T)lf) lo = mid + 1; else hi = mid; }   // uniform: every lane the same search
                            uint32_t qn; const uint32_t c = MaskPrefix<T>::rank((const T*)ablk, bc, rlen, lo, qn);
                            if (qn < rlen && (uint32_t)(rd(qn) - (T)lf) < ln) zbad = true;
                            if (stored) {
                                uint32_t rl2 = 0, rh2 = nres;                     // residuals below lf
                                while (rl2 < rh2) { const uint32_t mid = (rl2 + rh2) >> 1; if (resv(mid) < (T)lf) rl2 = mid + 1; else rh2 = mid; }
                                for (uint32_t jj = lane; jj < ln; jj += 64) wr(c + rl2 + before + jj, (T)(lf + jj));
                                wave_sync();
                                if (lane == 0) aiv[2 * kk] = c;                   // from here on the entry holds (kept elements before it, length)
                            }
                            before += ln;
                        }
                        wave_sync();
                        // ---- kept elements of the referenced list, 64 list positions per step (MaskedLongIterator.java:73-100)
                        uint32_t tbase = 0;                                       // kept elements before this step
                        for (uint32_t q0 = 0; q0 < rlen; q0 += 64) {
                            const uint32_t q = q0 + lane; const bool inl = q < rlen;
                            uint32_t lo = 0, hi = inl ? bc : 0u;                  // first block whose end position is > q
                            while (ballot(lo < hi)) { if (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (MaskPrefix<T>::pos((T)ablk[mid]) > q) hi = mid; else lo = mid + 1; } }
                            const bool keptq = inl && (lo < bc ? !(lo & 1u) : tailkeep);
                            const uint64_t km = ballot(keptq);
                            if (!km) continue;
                            const T v = keptq ? rd(q) : (T)0;
                            if (rep) chk += mix_node<T>(keptq ? k1 : 0u, v);
                            if (stored) {
                                const uint32_t t = tbase + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                                // extras in front of kept element t: residuals with c <= t (c non-decreasing: upper bound), intervals with c <= t
                                uint32_t l2 = 0, h2 = keptq ? nres : 0u;
                                while (ballot(l2 < h2)) { if (l2 < h2) { const uint32_t mid = (l2 + h2) >> 1; if (acr[mid] <= t) l2 = mid + 1; else h2 = mid; } }
                                uint32_t sh = l2;
                                for (uint32_t kk = 0; kk < ic; kk++) if (aiv[2 * kk] <= t) sh += aiv[2 * kk + 1];
                                if (keptq) wr(t + sh, v);
                            }
                            tbase += (uint32_t)__popcll(km);
                        }
                        if (tbase != kept) zbad = true;                           // the mask and the counts disagree
                    }
                    if (stored) {
                        if (lane == 0) { nd_base[(uint32_t)x & RM] = ob; nd_d[(uint32_t)x & RM] = d | (ogl ? 0x80000000u : 0u); }
                        if (ogl) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    }
                    wave_sync();
                }
            }
            if (ballot(zbad)) { failed = true; fail_need = 0xFFFFFFF5u; }
            blk_chk += chk;
        }

        err = wave_or32(err);
        if (failed) {
            if (lane == 0) {
                uint32_t slot = atomicAdd(a.fail_count, 1u);
                if (slot < a.fail_cap) { a.fail_list[slot] = bid; if (a.fail_need) a.fail_need[slot] = fail_need; }
            }
            continue;
        }
        blk_arcs = wave_sum64(blk_arcs); blk_chk = wave_sum64(blk_chk); blk_nodes = wave_sum64(blk_nodes);
        if (lane == 0) {
            unsigned long long* const accs = a.acc + (size_t)(bid & a.acc_mask) * kAccStride;
            atomicAdd(&accs[0], (unsigned long long)blk_arcs);
            atomicAdd(&accs[1], (unsigned long long)blk_chk);
            atomicAdd(&accs[2], (unsigned long long)blk_nodes);
            if (err) atomicOr(&accs[3], (unsigned long long)err);
        }
    }
}

}  // namespace

size_t flow_lds_bytes(uint32_t ring_cap) { return (size_t)ring_cap * sizeof(T) + (size_t)kFlowAux * 4; }

void launch_flow_scan(const DecodeArgs& a, uint32_t nblocks, uint32_t waves, void* scratch, uint32_t ring_cap, hipStream_t s) {
    if (nblocks == 0) return;
    const uint32_t grid = nblocks < waves ? nblocks : waves;
    hipLaunchKernelGGL(flow_kernel, dim3(grid), dim3(64), flow_lds_bytes(ring_cap), s, a, nblocks, (uint8_t*)scratch, (uint64_t)flow_scratch_bytes_per_wave(a.window), ring_cap, flow_glist_elems(a.window));
}

}  // namespace bvg
