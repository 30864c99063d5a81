// bvg_stream.hip — the fast decode kernel: a per-wavefront streaming data-flow machine.
//
// One wavefront owns one node block (plan: ~block_bits of compressed stream + its halo) and streams
// through it with everything it touches resident in LDS:
//
//   stream ring   4 KiB of the .graph bytes around the cursors, refilled in coalesced 1 KiB granules
//                 (16 B per lane) and byte-swapped once, so an MSB-first window is two/three ds_reads;
//   node ring     per-node state of the last 128 nodes: outdegree, list base, produced count, and the
//                 parsed HEADER of nodes not yet emitted (reference, #blocks, #intervals, #residuals,
//                 bit cursors of the interval and residual sections, checksum key);
//   block ring    the copy-block lengths of queued / in-flight nodes (decoded once by the parse row);
//   list ring     the successor lists of the nodes in flight and of the <= W nodes behind them.
//
// Two alternating activities:
//   parse row     (static, one node per lane) decodes the headers of the next 64 nodes: outdegree (gamma),
//                 reference (unary), block count + blocks (gamma, kept in the block ring), interval count,
//                 then SCANS the intervals only to find where the residuals start and how many there
//                 are (BVGraph.java:1010-1062), and queues the nodes;
//   emission      (dynamic) every lane is a worker: it grabs the next queued node IN ORDER, allocates its
//                 list in the list ring and emits one successor per iteration by the three-way merge of
//                 BVGraph.java:1062-1090 — masked copy of the referenced list (MaskedLongIterator),
//                 intervals and residual gaps, the latter two decoded LAZILY from the stream ring exactly
//                 when the merge consumes them (as the reference's ResidualLongIterator does).  A lane
//                 whose referenced list is still being produced waits on that node's `produced` counter,
//                 so reference chains pipeline; finished lanes re-grab in batches.
// Lanes stay busy regardless of how outdegrees are distributed inside the block.  The loop body is
// written select-style (no per-lane branches on the hot path) and reads the stream from LDS only.
//
// Anything this kernel cannot do in LDS — a list that does not fit the list ring, a record that does
// not fit the stream ring, codes longer than 64 bits, non-default codings (GEN) with exotic codes —
// makes the block FAIL OVER to the row-static kernel over global memory (bvg_kernels.hip, slow path).
#include "bvg_kernels.h"
#include "bvg_lds_codes.h"

namespace bvg {

namespace {

constexpr int NR = 128;                       // node ring entries
constexpr uint32_t NRM = NR - 1;
constexpr uint32_t kStreamWords = 1024;       // stream ring: 4 KiB
constexpr uint32_t SWM = kStreamWords - 1;
constexpr uint32_t kChunkBits = 8192;         // refill granule: 1 KiB
constexpr uint32_t kRingBits = kStreamWords * 32;
constexpr uint32_t kBlkRing = 512;            // block ring entries
constexpr uint32_t BRM = kBlkRing - 1;
constexpr uint32_t kInf = 0xFFFFFFFFu;


template <typename T> __device__ __forceinline__ T sentinel() { return (T)~(T)0; }

__device__ __forceinline__ uint32_t wave_incl_scan32(uint32_t v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if ((int)lane_id() >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_min32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ uint32_t wave_max32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}


__device__ __forceinline__ uint32_t win32(const uint32_t* r, uint32_t rel) { return bvg::win32<SWM>(r, rel); }
__device__ __forceinline__ uint64_t win64(const uint32_t* r, uint32_t rel) { return bvg::win64<SWM>(r, rel); }
__device__ __forceinline__ uint32_t decode_generic(const uint32_t* r, uint32_t rel, int coding, uint32_t k, uint64_t* out) { return decode_generic_w(bvg::win64<SWM>(r, rel), coding, k, out); }

template <typename T, bool MAT, bool GEN>
__global__ void __launch_bounds__(64) stream_kernel(DecodeArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t sring[kStreamWords];
    __shared__ uint32_t nd_d[NR], nd_base[NR], nd_prod[NR];
    __shared__ uint32_t m_refnb[NR], m_sb[NR], m_ni[NR], m_pi[NR], m_nr[NR], m_pr[NR], m_pe[NR], m_ps[NR], m_k0[NR], m_k1[NR];
    __shared__ uint32_t bring[kBlkRing];
    __shared__ uint64_t m_out[MAT ? NR : 1];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    T* const pool = reinterpret_cast<T*>(dyn_lds);
    const uint32_t CAP = a.lds_pool_elems, PM = CAP - 1;          // power of two

    const unsigned lane = threadIdx.x;
    const uint32_t bid = a.work_list ? a.work_list[blockIdx.x] : (a.blk_lo + blockIdx.x);
    const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
    if (e <= a.from || s >= a.to || s >= e) return;
    const uint32_t halo = a.blk_halo[bid];
    const uint64_t hmask = a.blk_mask[bid];
    const uint32_t W = (uint32_t)a.window;
    const int64_t hs = s - (int64_t)halo;
    if (e - hs > 0x7FFFFF00ll) {                                   // node ids are kept relative to hs in 32 bits
        if (lane == 0) { uint32_t slot = atomicAdd(a.fail_count, 1u); if (slot < a.fail_cap) a.fail_list[slot] = bid; }
        return;
    }
    const uint32_t nn = (uint32_t)(e - hs);                        // nodes are 0..nn-1 relative to hs
    const uint32_t s_r = (uint32_t)halo;                           // first block node (relative)
    const uint32_t rep_lo = (uint32_t)((s > a.from ? s : a.from) - hs), rep_hi = (uint32_t)((e < a.to ? e : a.to) - hs);
    const uint64_t sb0 = (a.offsets[hs] >> 7) << 7;                // stream base: 16-byte aligned, rel = abs - sb0
    const uint32_t zk = (uint32_t)a.cod.zeta_k;
    const uint32_t minint = (uint32_t)a.min_interval;
    const bool zfast = !GEN && zk >= 2;

    for (unsigned i = lane; i < (unsigned)NR; i += 64) { nd_d[i] = 0; nd_base[i] = 0; nd_prod[i] = 0; m_pe[i] = 0; }
    __syncthreads();

    // ---- wave-uniform state ----
    uint32_t parsed = 0, next = 0;           // [next, parsed) = parsed, not yet grabbed (relative node ids)
    uint32_t head = 0;                       // list ring allocation counter (index = counter & PM)
    uint32_t bhead = 0;                      // block ring allocation counter
    uint32_t whi = 0;                        // stream ring holds rel bits [whi - kRingBits, whi)
    bool failed = false;
    unsigned err = 0;
    uint64_t acc_arcs = 0, acc_chk = 0, acc_nodes = 0;

    // ---- per-lane worker state ----
    bool busy = false, rep = false;
    uint32_t x = 0;                                                 // relative node id
    T xT = 0;                                                       // absolute node id in T arithmetic
    uint32_t d = 0, j = 0, ob = 0, myslot = 0;
    uint32_t cb = 0, clen = 0, cpos = 0, keep = 0, nb = 0, sb = 0, cslot = 0; bool bfirst = false;
    T ivcur = 0, ivprev = 0; uint32_t ivrem = 0, ni = 0, pi = 0; bool ivfirst = false;
    T rhead = 0; uint32_t nr = 0, pr = 0, pe = 0; bool rvalid = false, rfirst = false, hadres = false;
    uint32_t k0 = 0, k1 = 0; uint64_t out0 = 0;

    for (;;) {
        // lowest incomplete node: everything below it is finished
        const uint32_t lowx = wave_min32(busy ? x : next);
        const uint32_t tailnode = lowx > W ? lowx - W : 0;                    // oldest node whose list may still be read
        // ================================================================ parse a row of headers
        bool did_parse = false;
        if (parsed < nn && parsed - next < 64 && parsed + 64 - tailnode <= (uint32_t)NR) {
            const uint32_t px = parsed + lane;
            const bool in_range = px < nn;
            const uint32_t hbit = px < s_r ? s_r - 1 - px : 0;
            const bool needed = in_range && (px >= s_r || ((hmask >> hbit) & 1ull));
            uint64_t off_x = 0, rec_end = 0;
            if (in_range) { off_x = a.offsets[hs + px]; rec_end = a.offsets[hs + px + 1]; }
            const uint32_t left = nn - parsed;
            const uint64_t row_hi = __shfl(rec_end, left >= 64 ? 63 : (int)left - 1, 64);
            if (row_hi - sb0 > 0xFFFF0000ull) { failed = true; break; }        // block span beyond 32-bit rel positions
            const uint32_t want = (uint32_t)(row_hi - sb0) + 96;               // windows read up to 96 bits past a cursor
            // the ring must keep every cursor of the nodes still in flight / queued: they start at m_pe[lowx-1]
            const uint32_t oldest = lowx < parsed ? m_ps[lowx & NRM] : (uint32_t)(__shfl(off_x, 0, 64) - sb0);
            const uint32_t new_whi = whi >= want ? whi : ((want + kChunkBits - 1) / kChunkBits) * kChunkBits;
            const bool span_ok = new_whi - (oldest & ~127u) <= kRingBits;
            if (!span_ok) {
                if (!ballot(busy) && parsed == next) { failed = true; break; }      // one row does not fit the ring: slow path
            } else {
                while (whi < new_whi) {                                         // coalesced 16 B / lane refill, byte-swapped
                    const uint64_t byte = (sb0 >> 3) + ((uint64_t)whi >> 3) + ((uint64_t)lane << 4);
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (byte + 16 <= a.padded_bytes) v = *reinterpret_cast<const uint4*>(a.graph + byte);
                    uint4 w; w.x = __builtin_bswap32(v.x); w.y = __builtin_bswap32(v.y); w.z = __builtin_bswap32(v.z); w.w = __builtin_bswap32(v.w);
                    *reinterpret_cast<uint4*>(&sring[((whi >> 5) + (lane << 2)) & SWM]) = w;
                    whi += kChunkBits;
                }
                __syncthreads();
                // ---- outdegree, reference, block count (all lanes in step)
                uint32_t rel = (uint32_t)(off_x - sb0);
                const uint32_t pend = (uint32_t)(rec_end - sb0);
                bool bad = false;
                uint32_t pd = 0, h_ref = 0, h_nb = 0;
                uint64_t v;
                if (needed) {
                    const uint32_t l = GEN ? decode_generic(sring, rel, a.cod.outdegree, 0, &v) : gamma64(win64(sring, rel), v);   // BVG:654-660
                    bad |= l == 0 || v > 0x7FFFFFFFull; rel += l; pd = bad ? 0u : (uint32_t)v;
                }
                const uint32_t slot = px & NRM;
                if (in_range) { nd_d[slot] = pd; nd_prod[slot] = 0; m_ps[slot] = (uint32_t)(off_x - sb0); }
                __syncthreads();
                if (pd > 0 && W > 0) {                                          // BVG:1015, readReference BVG:692-703
                    uint32_t l;
                    if (GEN) l = decode_generic(sring, rel, a.cod.reference, 0, &v);
                    else { const uint64_t w = win64(sring, rel); const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u; v = lz; l = lz < 64 ? lz + 1 : 0; }
                    bad |= l == 0; rel += l;
                    if (v > W || v > px) { err |= ERR_REF_RANGE; v = 0; }          // (a needed node never points before the halo)
                    h_ref = (uint32_t)v;
                }
                if (h_ref > 0) {                                                // readBlockCount, BVG:728-735
                    const uint32_t l = GEN ? decode_generic(sring, rel, a.cod.block_count, 0, &v) : gamma64(win64(sring, rel), v);
                    bad |= l == 0 || v > pend - rel + 1; rel += l; h_nb = bad ? 0u : (uint32_t)v;
                }
                // ---- block ring allocation for the whole row (uniform point), then the blocks
                const uint32_t bincl = wave_incl_scan32(h_nb);
                const uint32_t btot = __shfl(bincl, 63, 64);
                const uint32_t btail = lowx < parsed ? m_sb[lowx & NRM] : bhead;
                const bool bfits = btot <= kBlkRing - (bhead - btail);
                if (!bfits) {
                    if (!ballot(busy) && parsed == next) { failed = true; break; }  // a single row's blocks exceed the ring
                } else if (ballot(bad)) { failed = true; break; }
                else {
                    did_parse = true;
                    const uint32_t h_sb = bhead + bincl - h_nb;
                    int64_t extra = pd;
                    if (h_ref > 0) {                                            // BVG:1020-1032
                        int64_t copied = 0, tot = 0;
                        for (uint32_t i = 0; i < h_nb; i++) {
                            const uint32_t l = GEN ? decode_generic(sring, rel, a.cod.block, 0, &v) : gamma64(win64(sring, rel), v);
                            if (l == 0 || rel > pend) { bad = true; break; }
                            rel += l;
                            const uint32_t b = (uint32_t)v + (i ? 1u : 0u);
                            bring[(h_sb + i) & BRM] = b;
                            tot += b;
                            if (!(i & 1)) copied += b;
                        }
                        if (!(h_nb & 1)) copied += (int64_t)nd_d[(px - h_ref) & NRM] - tot;     // BVG:1030
                        extra = (int64_t)pd - copied;
                        if (extra < 0 || copied < 0) { err |= ERR_MALFORMED; extra = 0; }
                    }
                    uint32_t h_ni = 0, h_pi = 0;
                    if (extra > 0 && minint != 0) {                             // BVG:1037-1060: intervals are always gamma; scanned here
                        uint32_t l = gamma64(win64(sring, rel), v);
                        bad |= l == 0 || v > (pend - rel) / 2 + 1; rel += l; h_ni = bad ? 0u : (uint32_t)v;
                        h_pi = rel;
                        for (uint32_t i = 0; i < h_ni; i++) {
                            l = gamma64(win64(sring, rel), v);
                            if (l == 0 || rel > pend) { bad = true; break; }
                            rel += l;
                            l = gamma64(win64(sring, rel), v);
                            if (l == 0) { bad = true; break; }
                            rel += l;
                            extra -= (int64_t)v + minint;
                        }
                        if (extra < 0) { err |= ERR_MALFORMED; extra = 0; }
                    }
                    const uint32_t h_nr = (uint32_t)extra;
                    if (needed && pd > 0 && h_nr == 0 && rel != pend && !bad) err |= ERR_MALFORMED;
                    if (needed && pd == 0 && rel != pend && !bad) err |= ERR_MALFORMED;
                    if (in_range) {
                        m_refnb[slot] = h_ref | (h_nb << 8); m_sb[slot] = h_sb; m_ni[slot] = h_ni; m_pi[slot] = h_pi;
                        m_nr[slot] = h_nr; m_pr[slot] = rel; m_pe[slot] = pend;
                        const bool prep = px >= rep_lo && px < rep_hi;
                        if (!MAT) { uint32_t q0, q1; node_key((uint64_t)(hs + px) + a.node_base, q0, q1); m_k0[slot] = q0; m_k1[slot] = q1; }
                        else m_out[slot] = prep ? (a.batch ? a.cum[bid >> 1] : a.cum[hs + px - a.from]) : 0;
                    }
                    if (ballot(bad)) { failed = true; break; }
                    bhead += btot;
                    parsed += left >= 64 ? 64 : left;
                    __syncthreads();
                }
            }
        }

        // ================================================================ batched in-order grab
        unsigned grabbed = 0;
        {
            const uint64_t idlemask = ballot(!busy);
            const uint32_t qlen = parsed - next;
            if (idlemask && qlen > 0) {
                const uint32_t rank = (uint32_t)__popcll(idlemask & ((1ull << lane) - 1ull));
                const bool take = !busy && rank < qlen;
                const uint32_t cand = next + rank;
                const uint32_t cs = cand & NRM;
                const uint32_t cd = take ? nd_d[cs] : 0;
                const uint32_t incl = wave_incl_scan32(cd > CAP ? CAP + 1 : cd);
                const uint32_t tail = tailnode < next ? nd_base[tailnode & NRM] : head;
                const bool ok = take && incl <= CAP - (head - tail);
                grabbed = (unsigned)__popcll(ballot(ok));
                if (ok) {
                    x = cand; myslot = cs; d = cd; j = 0;
                    xT = (T)((uint64_t)(hs + cand));
                    ob = head + incl - cd;
                    nd_base[cs] = ob;
                    rep = cand >= rep_lo && cand < rep_hi;
                    busy = cd > 0;
                    const uint32_t refnb = m_refnb[cs];
                    const uint32_t ref = refnb & 0xFFu;
                    nb = refnb >> 8; sb = m_sb[cs]; ni = m_ni[cs]; pi = m_pi[cs]; nr = m_nr[cs]; pr = m_pr[cs]; pe = m_pe[cs];
                    hadres = nr > 0;
                    clen = 0; cpos = 0; keep = 0; cslot = cs; bfirst = true;
                    if (ref > 0) {
                        cslot = (cand - ref) & NRM;
                        cb = nd_base[cslot]; clen = nd_d[cslot];
                        if (nb == 0) keep = kInf;                                 // MaskedLongIterator.java:73-78
                    }
                    ivrem = 0; ivfirst = true; rvalid = false; rfirst = true;
                    if (rep) {
                        if (!MAT) { k0 = m_k0[cs]; k1 = m_k1[cs]; }
                        else { out0 = m_out[cs]; if (a.outdeg && !a.batch) a.outdeg[hs + cand - a.from] = (int32_t)cd; }
                        acc_arcs += cd; acc_nodes += 1;
                    }
                }
                head += wave_max32(ok ? incl : 0u);
                next += grabbed;
            }
        }
        if (!ballot(busy)) {
            if (next >= nn) break;                                                 // block finished
            if (!grabbed && !did_parse) { failed = true; break; }                  // the head of the queue can never fit: slow path
            continue;
        }

        // ================================================================ emission burst
        const uint32_t wlo = whi > kRingBits ? whi - kRingBits : 0;
        (void)wlo;
        for (;;) {
            // ---- copy stream: (re)load the mask state from the block ring (MaskedLongIterator.java:81-100)
            {
                const bool needblk = busy && cpos < clen && keep == 0;
                if (ballot(needblk)) {
                    if (needblk) {
                        if (!bfirst) {                                             // a keep block just ended: skip block follows
                            if (nb == 0) cpos = clen;                              // odd count: the tail is dropped
                            else { cpos += bring[sb & BRM]; sb++; nb--; }
                        }
                        bfirst = false;
                        if (cpos < clen) {
                            if (nb == 0) keep = kInf;                              // even count: the tail is kept
                            else { keep = bring[sb & BRM]; sb++; nb--; }           // may be 0 only for the very first block
                        }
                    }
                }
            }
            // ---- interval stream: decode the next (left, len) pair lazily (BVG:1047-1056)
            {
                const bool needint = busy && ivrem == 0 && ni > 0;
                if (ballot(needint)) {
                    if (needint) {
                        uint64_t v1, v2;
                        const uint32_t l1 = gamma64(win64(sring, pi), v1);
                        const uint32_t l2 = gamma64(win64(sring, pi + l1), v2);
                        if (l1 == 0 || l2 == 0) err |= ERR_CAPACITY;               // > 63-bit gamma: fail over to the slow path
                        pi += l1 + l2;
                        ivcur = ivfirst ? (T)(xT + (T)nat2int64(v1)) : (T)(ivprev + 1 + (T)v1);
                        ivrem = (uint32_t)v2 + minint;
                        ivprev = ivcur + (T)ivrem;
                        ivfirst = false; ni--;
                    }
                }
            }
            // ---- residual stream: decode the next gap lazily (ResidualLongIterator, BVG:917,929)
            {
                const bool needres = busy && !rvalid && nr > 0;
                if (ballot(needres)) {
                    uint64_t val = 0; uint32_t len = 0;
                    if (zfast) {                                                   // zeta_k from a 32-bit window, branch-free
                        const uint32_t w = win32(sring, pr);
                        const uint32_t z = w ? (uint32_t)__builtin_clz(w) : 32u;
                        const uint32_t nbz = z * zk + zk - 1, zt = z + 1 + nbz;
                        const uint32_t t = (w << ((z + 1) & 31u)) >> ((32u - nbz) & 31u);
                        const uint32_t leftv = 1u << ((z * zk) & 31u);
                        const bool lt = t < leftv;
                        const uint32_t v1 = ((t << 1) | ((w >> ((31u - zt) & 31u)) & 1u)) - 1u;
                        val = lt ? t + leftv - 1u : v1;
                        len = zt < 32 ? zt + (lt ? 0u : 1u) : 0u;
                    }
                    if (ballot(needres && len == 0)) {                             // long code / other coding: 64-bit window
                        if (needres && len == 0) {
                            if (GEN) len = decode_generic(sring, pr, a.cod.residual, zk, &val);
                            else len = zeta64(win64(sring, pr), zk, val);
                            if (len == 0) err |= ERR_CAPACITY;
                        }
                    }
                    if (needres) {
                        pr += len;
                        rhead = rfirst ? (T)(xT + (T)nat2int64(val)) : (T)(rhead + 1 + (T)val);
                        rfirst = false; rvalid = true; nr--;
                    }
                }
            }
            // ---- merge step: emit one successor (MergedLongIterator.java:63-92, three-way)
            {
                const bool chas = busy && cpos < clen && keep != 0;
                const bool blocked = (busy && cpos < clen && keep == 0) || (chas && __hip_atomic_load(&nd_prod[cslot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) <= cpos);
                const bool can = busy && !blocked && (rvalid || nr == 0) && (ivrem != 0 || ni == 0);
                const T c = chas ? pool[(cb + cpos) & PM] : sentinel<T>();
                const T iv = ivrem ? ivcur : sentinel<T>();
                const T r = rvalid ? rhead : sentinel<T>();
                T m = c < iv ? c : iv; m = m < r ? m : r;
                if (can) {
                    pool[(ob + j) & PM] = m;
                    if (rep) {
                        const uint64_t y64 = m == sentinel<T>() ? ~0ull : (uint64_t)m + a.node_base;
                        if (!MAT) acc_chk += mix_keyed(k0, k1, y64);
                        else a.succ[out0 + j] = m == sentinel<T>() ? -1ll : (int64_t)y64;
                    }
                    j++;
                    __hip_atomic_store(&nd_prod[myslot], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const bool ec = chas && c == m, ei = ivrem != 0 && iv == m, er = rvalid && r == m;
                    cpos += ec ? 1u : 0u; keep -= ec ? 1u : 0u;
                    ivcur += ei ? 1 : 0; ivrem -= ei ? 1u : 0u;
                    rvalid = rvalid && !er;
                    if (j == d) {
                        busy = false;
                        if (hadres && nr == 0 && pr != pe) err |= ERR_MALFORMED;   // the record must end where the next one starts
                    }
                }
            }
            const uint64_t bm = ballot(busy);
            if (!bm) break;
            if ((unsigned)__popcll(~bm) >= a.grab_threshold && (parsed > next || parsed < nn)) break;
        }
    }

    err = wave_or32(err);
    if (failed || (err & ERR_CAPACITY)) {
        if (lane == 0) {
            uint32_t slot = atomicAdd(a.fail_count, 1u);
            if (slot < a.fail_cap) a.fail_list[slot] = bid;
        }
        return;
    }
    acc_arcs = wave_sum64(acc_arcs); acc_chk = wave_sum64(acc_chk); acc_nodes = wave_sum64(acc_nodes);
    if (lane == 0) {
        unsigned long long* const accs = a.acc + (size_t)(bid & a.acc_mask) * kAccStride;   // this block's result stripe
        atomicAdd(&accs[0], (unsigned long long)acc_arcs);
        atomicAdd(&accs[1], (unsigned long long)acc_chk);
        atomicAdd(&accs[2], (unsigned long long)acc_nodes);
        if (err) atomicOr(&accs[3], (unsigned long long)err);
    }
}


}  // namespace

void launch_stream_decode(const DecodeArgs& a, uint32_t nblocks, bool wide, bool materialise, hipStream_t s) {
    if (nblocks == 0) return;
    dim3 grid(nblocks), block(64);
    const bool gen = !(a.cod.outdegree == BVG_GAMMA && a.cod.reference == BVG_UNARY && a.cod.block_count == BVG_GAMMA &&
                       a.cod.block == BVG_GAMMA && a.cod.residual == BVG_ZETA);
    const size_t dyn = (size_t)a.lds_pool_elems * (wide ? 8 : 4);
#define BVG_SL(T, M) do { if (gen) hipLaunchKernelGGL((stream_kernel<T, M, true>), grid, block, dyn, s, a); \
                          else hipLaunchKernelGGL((stream_kernel<T, M, false>), grid, block, dyn, s, a); } while (0)
    if (!wide) { if (!materialise) BVG_SL(uint32_t, false); else BVG_SL(uint32_t, true); }
    else { if (!materialise) BVG_SL(uint64_t, false); else BVG_SL(uint64_t, true); }
#undef BVG_SL
}

}  // namespace bvg
