// bvg_index.hip — the FILLING pass of the residual skip index as a dense walk (round 4).
//
// The skip index (bvg_api.hip build_skip) holds, for every list of >= kSkipMin residuals, one entry per kSkipEvery residuals:
// {bit offset of that residual's code from the record start, value of the residual before it}.  Finding those offsets means WALKING the
// list's codes one after the other (a code's length is known only by decoding it: BVG:787-796, 902-935).  Rounds 1-3 did that inside the
// validating decode of the row kernel, one lane per node of a row: the longest list of a row (~270 residuals on the default workload,
// where the mean is 30) set the step count of the whole row with a handful of lanes busy -- a third of a 2 s pass.
//
// Here the walk is its own kernel and it is DENSE: one wavefront per node block parses the record headers of the block row by row
// (one lane per record: outdegree, reference, copy blocks and intervals only as far as their sums go -- BVG:1010-1060 -- which gives the
// residual count and where the residual codes start), QUEUES the lists that have entries, and whenever 64 of them are waiting (or the
// block ends) walks them with one lane per list: every lane of the walk is a long list.  The block's stream is staged in LDS once (8 KiB window;
// what lies outside it is read from global memory) and read through the generic BitCursor (every legal coding), nothing is emitted, summed or validated: the entries are verified
// by the pass that follows (the row kernel decoding WITH them, skip_mode 3: an entry that does not lie where the stream says fails its
// task and leaves the block unvalidated), so a wrong walk can cost speed, never correctness.
//
// The entries land exactly where the counting pass (skip_mode 1) allotted them: node order within the block, a.skip_first[bid] onwards.
#include "bvg_kernels.h"

namespace bvg {

namespace {

constexpr uint32_t RM = kRing - 1;
constexpr uint32_t kWalkWords = 2048;          // LDS window over the block's stream: 8 KiB (a block is ~4 KiB of stream + its halo)

template <bool GEN> struct Rdi {   // the field readers of bvg_kernels.hip (GEN = false: BVGraph's default codings, straight-line)
    static __device__ __forceinline__ uint64_t outdegree(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.outdegree, 0, g); return c.read_gamma(g); }
    static __device__ __forceinline__ uint64_t reference(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.reference, 0, g); return c.read_unary(g); }
    static __device__ __forceinline__ uint64_t block_count(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.block_count, 0, g); return c.read_gamma(g); }
    static __device__ __forceinline__ uint64_t block(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.block, 0, g); return c.read_gamma(g); }
    static __device__ __forceinline__ uint64_t residual(BitCursor& c, const Codings& k, uint64_t g) { if (GEN) return c.read_coded(k.residual, (unsigned)k.zeta_k, g); return c.read_zeta((unsigned)k.zeta_k, g); }
};

// T: the type of the entries' values (uint32_t, or uint64_t for an index built by the 64-bit kernels)
template <typename T, bool GEN>
__global__ void __launch_bounds__(64) index_walk_kernel(DecodeArgs a) {
    typedef Rdi<GEN> R;
    __shared__ uint32_t nd_d[kRing];
    __shared__ uint32_t q_lo[64], q_hi[64], q_rec[64], q_nres[64], q_ent[64], q_x[64], q_len[64];   // the waiting lists: start of the residual codes (bit), its distance
                                                                                        // from the record start, residuals, first entry, node (relative to hs)
    const unsigned lane = threadIdx.x;
    const uint32_t kSkipMin = a.skip_min, kSkipShift = a.skip_shift, kSkipEvery = 1u << kSkipShift;   // (this index's granularity: they hide the compile-time defaults of bvg_kernels.h)
    const uint32_t bid = a.work_list ? a.work_list[blockIdx.x] : (a.blk_lo + blockIdx.x);
    const int64_t s = (int64_t)a.blk_first[bid], e = (int64_t)a.blk_first[bid + 1];
    if (e <= a.from || s >= a.to || s >= e) return;
    const uint64_t sk_base = a.skip_first[bid];
    const uint32_t sk_n = (uint32_t)(a.skip_first[bid + 1] - sk_base);
    if (sk_n == 0) return;                                              // no list of this block has entries
    const uint32_t halo = a.blk_halo[bid];
    const uint64_t hmask = a.blk_mask[bid];
    const int W = a.window;
    const int64_t hs = s - (int64_t)halo;
    for (unsigned i = lane; i < (unsigned)kRing; i += 64) nd_d[i] = 0;
    // the block's stream (halo included) in LDS, big-endian dwords: a block is ~4 KiB; what does not fit the window is read from global memory (BitCursor::peek)
    __shared__ __attribute__((aligned(16))) uint32_t win[kWalkWords];
    uint64_t win_bit0 = 0; uint32_t win_bits = 0;
    {
        const uint64_t lo_bit = a.offsets[hs], hi_bit = a.offsets[e];
        const uint64_t b0 = (lo_bit >> 3) & ~15ull;
        uint64_t nb = ((hi_bit + 7) >> 3) + 16 > b0 ? ((hi_bit + 7) >> 3) + 16 - b0 : 0;          // (+16: a window read looks 96 bits ahead)
        nb = (nb + 15) & ~15ull;
        if (nb > kWalkWords * 4ull) nb = kWalkWords * 4ull;
        if (b0 + nb > a.padded_bytes) nb = a.padded_bytes > b0 ? (a.padded_bytes - b0) & ~15ull : 0;
        for (uint32_t c = lane; c < (uint32_t)(nb >> 4); c += 64) {
            const uint4 v = *reinterpret_cast<const uint4*>(a.graph + b0 + ((uint64_t)c << 4));
            uint4 w; w.x = __builtin_bswap32(v.x); w.y = __builtin_bswap32(v.y); w.z = __builtin_bswap32(v.z); w.w = __builtin_bswap32(v.w);
            *reinterpret_cast<uint4*>(&win[c << 2]) = w;
        }
        win_bit0 = b0 << 3; win_bits = (uint32_t)(nb << 3);
    }
    __syncthreads();

    uint32_t sk_run = 0, qn = 0;
    auto walk = [&]() {                                                 // one lane per waiting list
        const bool on = lane < qn;
        const uint64_t pos0 = on ? (((uint64_t)q_hi[lane] << 32) | q_lo[lane]) : 0ull;
        const uint32_t nres = on ? q_nres[lane] : 0u, ent = on ? q_ent[lane] : 0u, recoff = on ? q_rec[lane] : 0u;
        const int64_t x = hs + (int64_t)(on ? q_x[lane] : 0u);
        uint32_t nmax = nres;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = __shfl_xor(nmax, o, 64); nmax = t > nmax ? t : nmax; }
        BitCursor cur{a.graph, pos0, a.limit_byte, win, win_bit0, win_bits, 0u, 0xFFFFFFFFu};
        const uint64_t guard = pos0 + (on ? q_len[lane] : 0u);          // the end of the list's record: a runaway code stops there
        int64_t r = x;
        for (uint32_t t = 0; t < nmax; t++) {
            if (t < nres) {
                if (t && (t & (kSkipEvery - 1u)) == 0) {                // the entry of this residual: where its code starts, what came before it
                    const uint32_t ei = ent + (t >> kSkipShift) - 1u;
                    if (ei < sk_n) {
                        const uint64_t rel = cur.pos - pos0 + recoff;
                        a.skip_bit[sk_base + ei] = (uint16_t)(rel < 0xFFFFull ? rel : 0xFFFFull);      // (0xFFFF: unusable, the reader fails over)
                        reinterpret_cast<T*>(a.skip_val)[sk_base + ei] = (T)(uint64_t)r;
                    }
                }
                const uint64_t v = R::residual(cur, a.cod, guard);
                r = t == 0 ? r + nat2int(v) : r + 1 + (int64_t)v;
            }
        }
        __syncthreads();
        qn = 0;
    };

    for (int64_t r0 = hs; r0 < e; r0 += 64) {
        const int64_t x = r0 + lane;
        const bool in_range = x < e;
        const uint64_t hbit = x < s ? (uint64_t)(s - 1 - x) & 63u : 0;
        const bool needed = in_range && (x >= s || ((hmask >> hbit) & 1ull));
        uint64_t off_x = 0, rec_end = 0;
        if (needed) { off_x = a.offsets[x]; rec_end = a.offsets[x + 1]; }
        BitCursor cur{a.graph, off_x, a.limit_byte, win, win_bit0, win_bits, 0u, 0xFFFFFFFFu};
        uint32_t d = 0;
        bool ok = needed;
        if (needed) { const uint64_t dv = R::outdegree(cur, a.cod, rec_end); ok = dv <= 0x7FFFFFFFull; d = ok ? (uint32_t)dv : 0u; }
        if (needed) nd_d[(uint32_t)x & RM] = d;
        __syncthreads();
        uint32_t nres = 0;
        if (needed && d > 0) {
            uint32_t ref = 0;
            if (W > 0) { const uint64_t rv = R::reference(cur, a.cod, rec_end); if (rv > (uint64_t)W || (int64_t)rv > x) ok = false; else ref = (uint32_t)rv; }
            int64_t extra = d;
            if (ref > 0) {
                uint64_t nb = R::block_count(cur, a.cod, rec_end);
                if (nb > rec_end - (cur.pos < rec_end ? cur.pos : rec_end) + 1) { ok = false; nb = 0; }
                int64_t copied = 0, tot = 0;
                for (uint64_t i = 0; i < nb; i++) {
                    const int64_t b = (int64_t)R::block(cur, a.cod, rec_end) + (i ? 1 : 0);
                    tot += b; if (!(i & 1)) copied += b;
                    if (cur.pos > rec_end) { ok = false; break; }
                }
                if (!(nb & 1)) copied += (int64_t)nd_d[(uint32_t)(x - ref) & RM] - tot;           // BVG:1030
                extra = (int64_t)d - copied;
                if (extra < 0 || copied < 0) { ok = false; extra = 0; }
            }
            if (extra > 0 && a.min_interval != 0) {                                              // always gamma (BVG:1040-1058)
                uint64_t ni = cur.read_gamma(rec_end);
                if (ni > (rec_end - (cur.pos < rec_end ? cur.pos : rec_end)) / 2 + 1) { ok = false; ni = 0; }
                for (uint64_t i = 0; i < ni; i++) {
                    (void)cur.read_gamma(rec_end);
                    extra -= (int64_t)cur.read_gamma(rec_end) + a.min_interval;
                    if (cur.pos > rec_end) { ok = false; break; }
                }
                if (extra < 0) { ok = false; extra = 0; }
            }
            nres = ok ? (uint32_t)extra : 0u;
        }
        const uint32_t cntE = (needed && d > 0 && ok && nres >= kSkipMin) ? (nres - 1u) >> kSkipShift : 0u;
        uint32_t eincl = cntE;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(eincl, o, 64); if ((int)lane >= o) eincl += t; }
        const uint32_t efirst = sk_run + eincl - cntE;
        sk_run += __shfl(eincl, 63, 64);
        const uint64_t m = ballot(cntE != 0);
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (qn + cnt > 64u) walk();                                     // (wave-uniform)
        if (cntE != 0) {
            const uint32_t slot = qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            q_lo[slot] = (uint32_t)cur.pos; q_hi[slot] = (uint32_t)(cur.pos >> 32); q_rec[slot] = (uint32_t)(cur.pos - off_x);
            q_nres[slot] = nres; q_ent[slot] = efirst; q_x[slot] = (uint32_t)(x - hs); q_len[slot] = (uint32_t)(rec_end > cur.pos ? rec_end - cur.pos : 0);
        }
        qn += cnt;
        __syncthreads();
    }
    if (qn) walk();
}

// Entries of blocks that no kernel finished with a mark (fmt 0: they ended in the generic kernel, which uses none) are never read; the dense walk may have
// written some before the block failed over.  They are cleared, so that an index -- and its file -- does not depend on how it was built.
__global__ void __launch_bounds__(256) clear_unmarked_entries_kernel(const uint64_t* first, const uint8_t* fmt, uint16_t* bit, void* val, uint32_t blo, uint32_t bhi, int wide) {
    const uint32_t b = blo + blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bhi || fmt[b] != 0) return;
    for (uint64_t i = first[b]; i < first[b + 1]; i++) { bit[i] = 0; if (wide) reinterpret_cast<uint64_t*>(val)[i] = 0; else reinterpret_cast<uint32_t*>(val)[i] = 0; }
}

}  // namespace

void launch_clear_unmarked_entries(const uint64_t* first, const uint8_t* fmt, uint16_t* bit, void* val, uint32_t blo, uint32_t bhi, bool wide, hipStream_t s) {
    if (bhi <= blo) return;
    hipLaunchKernelGGL(clear_unmarked_entries_kernel, dim3((bhi - blo + 255) / 256), dim3(256), 0, s, first, fmt, bit, val, blo, bhi, wide ? 1 : 0);
}

// fills the entries of the blocks of the work list (or of blk_lo + [0, nblocks)); `wide`: 64-bit entry values
void launch_index_walk(const DecodeArgs& a, uint32_t nblocks, bool wide, hipStream_t s) {
    if (nblocks == 0) return;
    const bool gen = !(a.cod.outdegree == BVG_GAMMA && a.cod.reference == BVG_UNARY && a.cod.block_count == BVG_GAMMA && a.cod.block == BVG_GAMMA && a.cod.residual == BVG_ZETA);
    const dim3 grid(nblocks), block(64);
    if (wide) { if (gen) hipLaunchKernelGGL((index_walk_kernel<uint64_t, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((index_walk_kernel<uint64_t, false>), grid, block, 0, s, a); }
    else { if (gen) hipLaunchKernelGGL((index_walk_kernel<uint32_t, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((index_walk_kernel<uint32_t, false>), grid, block, 0, s, a); }
}

}  // namespace bvg
