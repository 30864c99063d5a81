// bvg_derive.hip — the offsets index from a bare .graph, in parallel (BVGraph -O / writeOffsets, BVGraph.java:2595-2609;
// loadSequential / loadOffline, BVG:1345-1464).
//
// A record's length is only known by parsing it, and parsing it needs the outdegree of the node it references (BVG:1030), so the
// reference walks the stream once, sequentially.  Here the stream is cut into CHUNKS of equal bit length and every chunk is
// walked by its own lane.  A walk is a STATE MACHINE that decodes exactly one instantaneous code per step -- which field of the record
// it is in (outdegree, reference, block count, a block, interval count, interval left / length, a residual) and the counters that say
// what comes next (BVG:1003-1064) are the state, so the 64 lanes of a wavefront stay in step although they sit in 64 different
// records.  Chunk 0 starts in the true state; every other chunk starts from a GUESS ("a record starts at my first bit, the window
// before it is empty").  The walks are repeated, each chunk entering in the state its predecessor left in, until no exit state
// changes any more: after round r the chunks 0..r are exact by induction, so the iteration ends at the one true walk -- and in
// practice after a handful of rounds, because a walk that starts inside a record falls into step with the true record boundaries
// within its own chunk (the codes are instantaneous and records are short); only chunks whose entry state changed are walked
// again, so a region that cannot fall into step by itself (the inside of a record of millions of residuals) costs one
// single-chunk launch per chunk.  A last pass numbers the record starts (prefix sum of the per-chunk counts) and writes the
// offsets.  Anything odd in the final, exact walk (a code longer than 64 bits, a negative count, a reference beyond the window,
// too few records) is reported, and the caller falls back to the sequential walk, whose error bits are the documented ones.
#include "bvg_kernels.h"
#include "bvg_lds_codes.h"

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>

namespace bvg {

namespace {

constexpr int kMaxDeriveWindow = 127;          // outdegree ring per lane in LDS (larger windows take the sequential kernel)
constexpr uint32_t kChunkBits = 32768;         // 4 KiB of stream per lane
constexpr uint32_t kWarmChunks = 8;            // a guessed walk starts this many chunks before its own: by the time it enters it, it has almost always fallen into step
constexpr uint32_t kCrawlBelow = 2048;         // rounds with at most this many chunks to walk give each chunk a workgroup (derive_crawl_kernel)

enum : uint32_t { F_OUTDEG = 0, F_REF, F_BCOUNT, F_BLOCK, F_ICOUNT, F_ILEFT, F_ILEN, F_RES };

struct WalkState {                             // everything the walk needs to go on; fields a state does not use are kept 0 (states are compared)
    uint64_t pos;                              // bit position of the next code
    int64_t extra;                             // successors not yet accounted for (F_ICOUNT..F_ILEN) / residuals left (F_RES)
    int64_t copied, total;                     // F_BLOCK
    uint32_t field, d, ref, bc, bi, ic, ii, pad;
};

__device__ __forceinline__ bool same_state(const WalkState& a, const WalkState& b) {
    return a.pos == b.pos && a.extra == b.extra && a.copied == b.copied && a.total == b.total && a.field == b.field && a.d == b.d && a.ref == b.ref &&
           a.bc == b.bc && a.bi == b.bi && a.ic == b.ic && a.ii == b.ii;
}

// A 192-bit register buffer over the stream: aligned 8-byte loads, one per 64 bits consumed and issued one word ahead, instead of two
// dependent loads per code (a lone lane crawling through a region that cannot be guessed is bound by exactly that latency).
struct BitBuf {
    const uint64_t* words; uint64_t base; uint64_t w0, w1, w2;            // w0 = bits [base, base + 64), w1, w2 the next two words
    __device__ __forceinline__ void init(const uint8_t* g, uint64_t pos) {
        words = reinterpret_cast<const uint64_t*>(g); base = pos & ~63ull;
        const uint64_t i = base >> 6;
        w0 = __builtin_bswap64(words[i]); w1 = __builtin_bswap64(words[i + 1]); w2 = __builtin_bswap64(words[i + 2]);
    }
    __device__ __forceinline__ uint64_t peek(uint64_t pos) {
        while (pos - base >= 64) { w0 = w1; w1 = w2; base += 64; w2 = __builtin_bswap64(words[(base >> 6) + 2]); }   // (a code is at most 64 bits: one step almost always)
        const unsigned off = (unsigned)(pos - base);
        return off ? (w0 << off) | (w1 >> (64u - off)) : w0;
    }
};

// one code of the given coding from a 64-bit window: length (0 = longer than the window: impossible in a well-formed stream) and value
template <bool GEN>
__device__ __forceinline__ uint32_t read_code(uint64_t w, int coding, uint32_t k, uint64_t& v) {
    if (!GEN) {                                                          // default codings: gamma / unary / zeta_k, computed side by side
        const uint32_t lz = w ? (uint32_t)__builtin_clzll(w) : 64u;
        if (coding == BVG_UNARY) { v = lz; return lz < 64 ? lz + 1 : 0u; }
        if (coding == BVG_GAMMA) { const uint32_t len = 2 * lz + 1; v = lz < 32 ? (w >> (64u - len)) - 1 : 0; return lz < 32 ? len : 0u; }
        return zeta64(w, k, v);
    }
    return decode_generic_w(w, coding, k, &v);
}

// The walk of one chunk: from state `st` until the next code lies at or behind `end`.  ring = this lane's last W + 1 outdegrees in LDS
// (slot (j & M) * 64 of node j), cnt = nodes started so far (>= M + 1 when the walk begins).  emit(p) is called at every record start.
template <bool GEN, typename Emit>
__device__ __forceinline__ void walk(const uint8_t* g, WalkState& st, uint64_t end, uint32_t* ring, uint32_t M, uint32_t& cnt, int W, uint32_t minint, const Codings& cod,
                                     unsigned& bad, Emit emit, const uint8_t* lens = nullptr, uint64_t lens_lo = 0) {
    BitBuf bb; bb.init(g, st.pos);
    while (st.pos < end) {
        const uint32_t f = st.field;
        if (lens && f == F_RES) {                                          // a run of residuals: follow the table of code lengths (derive_crawl_kernel)
            while (st.extra > 0 && st.pos < end) { uint32_t l = lens[st.pos - lens_lo]; if (l == 0) { bad |= 1u; l = 64; } st.pos += l; st.extra--; }
            if (st.extra <= 0) { st.field = F_OUTDEG; st.extra = 0; st.d = 0; }
            continue;
        }
        if (f == F_OUTDEG) emit(st.pos);                                   // a record starts here
        const uint64_t w = bb.peek(st.pos);
        const int coding = f == F_OUTDEG ? cod.outdegree : f == F_REF ? cod.reference : f == F_BCOUNT ? cod.block_count : f == F_BLOCK ? cod.block : f == F_RES ? cod.residual : BVG_GAMMA;
        uint64_t v;
        uint32_t len = read_code<GEN>(w, coding, (uint32_t)cod.zeta_k, v);
        if (len == 0) { bad |= 1u; len = 64; v = 0; }                     // (a guessed walk may read anything; the exact walk must not)
        st.pos += len;
        bool to_extras = false, to_res = false;
        switch (f) {
            case F_OUTDEG: {
                if (v > 0x7FFFFFFFull) { bad |= 2u; v = 0x7FFFFFFFull; }
                st.d = (uint32_t)v; ring[(cnt & M) * 64] = st.d; cnt++;
                if (st.d == 0) { st.d = 0; }                              // empty list: the next record starts here
                else if (W > 0) st.field = F_REF;
                else { st.extra = st.d; to_extras = true; }
                break;
            }
            case F_REF: {
                if (v > (uint64_t)W) { bad |= 4u; v = 0; }
                st.ref = (uint32_t)v;
                if (st.ref) st.field = F_BCOUNT; else { st.extra = st.d; to_extras = true; }
                break;
            }
            case F_BCOUNT: {
                if (v > 0x7FFFFFFFull) { bad |= 2u; v = 0; }
                st.bc = (uint32_t)v; st.bi = 0; st.copied = 0; st.total = 0;
                if (st.bc) st.field = F_BLOCK;
                else { st.extra = (int64_t)st.d - (int64_t)ring[((cnt - 1 - st.ref) & M) * 64]; to_extras = true; }   // no blocks: everything is copied (BVG:1030)
                break;
            }
            case F_BLOCK: {
                const int64_t b = (int64_t)v + (st.bi ? 1 : 0);
                st.total += b; if (!(st.bi & 1u)) st.copied += b;
                if (++st.bi == st.bc) {
                    if (!(st.bc & 1u)) st.copied += (int64_t)ring[((cnt - 1 - st.ref) & M) * 64] - st.total;      // BVG:1030
                    st.extra = (int64_t)st.d - st.copied; to_extras = true;
                }
                break;
            }
            case F_ICOUNT: {
                if (v > 0x7FFFFFFFull) { bad |= 2u; v = 0; }
                st.ic = (uint32_t)v; st.ii = 0;
                if (st.ic) st.field = F_ILEFT; else to_res = true;
                break;
            }
            case F_ILEFT: st.field = F_ILEN; break;
            case F_ILEN: {
                st.extra -= (int64_t)v + (int64_t)minint;
                if (++st.ii == st.ic) to_res = true; else st.field = F_ILEFT;
                break;
            }
            default: {                                                    // F_RES
                if (--st.extra <= 0) { st.field = F_OUTDEG; st.extra = 0; }
                break;
            }
        }
        if (to_extras) {                                                  // the record's header is done: intervals, then residuals (BVG:1038-1064)
            st.ref = 0; st.bc = 0; st.bi = 0; st.copied = 0; st.total = 0;
            if (st.extra < 0) { bad |= 8u; st.extra = 0; }
            if (st.extra > 0 && minint != 0) st.field = F_ICOUNT; else to_res = true;
        }
        if (to_res) {
            st.ic = 0; st.ii = 0; st.d = 0;
            if (st.extra < 0) { bad |= 8u; st.extra = 0; }
            st.field = st.extra > 0 ? F_RES : F_OUTDEG;
        }
        if (st.field == F_OUTDEG) { st.d = 0; st.extra = 0; }
    }
}

// A state from which no record start can be expected within about a chunk: a run of thousands of residuals / blocks / intervals still to
// go.  True inside a giant record -- and the signature of a guessed walk that took garbage for an outdegree (a gamma code read from
// random bits is "a million" once in a million times, and then the walk skips a million codes blindly).  Such a state is handed to
// the next chunk only when it is EXACT; otherwise it would run through the following chunks one per round, replacing good guesses.
__device__ __forceinline__ bool blind_state(const WalkState& s) {
    return (s.field == F_RES && s.extra > 4096) || (s.field == F_BLOCK && s.bc - s.bi > 4096u) || ((s.field == F_ILEFT || s.field == F_ILEN) && s.ic - s.ii > 2048u);
}

struct RoundArgs {
    const uint8_t* g; uint64_t total_bits; uint32_t nchunks; int W; uint32_t minint; Codings cod;
    const uint32_t* list; uint32_t nlist;
    WalkState* entry; uint32_t* entry_ring; WalkState* exit; uint32_t* exit_ring; uint32_t* counts;
    uint32_t first_round;
};

// The walk of chunk c from entry[c] (first round: from the guess, see below) to its end: exit[c], counts[c].
template <bool GEN>
__device__ __forceinline__ void round_chunk(const RoundArgs& a, uint32_t c, uint32_t* ring, const uint8_t* lens) {
    const uint32_t R = (uint32_t)a.W + 1u;
    uint32_t M = 1; while (M < R) M <<= 1; M -= 1;
    WalkState st = a.entry[c];
    uint32_t* const my_ring = a.entry_ring + (size_t)c * R;
    // ring: my_ring[j] = outdegree of the node j + 1 places before the next one to start; local numbering starts at cnt = M + 1
    uint32_t cnt = M + 1;
    for (uint32_t j = 0; j < R; j++) ring[((cnt - 1 - j) & M) * 64] = my_ring[j];
    const uint64_t lo = (uint64_t)c * kChunkBits, hi = lo + kChunkBits < a.total_bits ? lo + kChunkBits : a.total_bits;
    unsigned bad = 0;
    uint32_t starts = 0;
    if (a.first_round && c > 0) {
        // The guess: a record starts kWarmChunks chunks before mine (at bit 0, the truth, for the first ones) with an empty window.  A
        // walk that starts inside a record falls into step with the true boundaries after some thousands of codes (it has to END a
        // bogus record exactly on a true start, and then parse a window's worth of records right): walking that distance first makes
        // the state in which it enters its own chunk -- and with it almost every exit of the first round -- the true one.
        st.pos = c > kWarmChunks ? lo - (uint64_t)kWarmChunks * kChunkBits : 0;
        walk<GEN>(a.g, st, lo, ring, M, cnt, a.W, a.minint, a.cod, bad, [&](uint64_t) {});
        a.entry[c] = st;
        for (uint32_t j = 0; j < R; j++) my_ring[j] = ring[((cnt - 1 - j) & M) * 64];
        bad = 0;
    }
    walk<GEN>(a.g, st, hi, ring, M, cnt, a.W, a.minint, a.cod, bad, [&](uint64_t) { starts++; }, lens, lo);
    if (c + 1 == a.nchunks && st.field == F_OUTDEG) starts++;             // the position behind the last record counts as a start too (offsets[n])
    a.counts[c] = starts;
    a.exit[c] = st;
    uint32_t* const xr = a.exit_ring + (size_t)c * R;
    for (uint32_t j = 0; j < R; j++) xr[j] = ring[((cnt - 1 - j) & M) * 64];
}

// Walk kernel, many chunks: a lane per chunk.
template <bool GEN>
__global__ void __launch_bounds__(64) derive_round_kernel(RoundArgs a) {
    extern __shared__ uint32_t ring_lds[];                                // (M + 1) slots x 64 lanes, M + 1 = the power of two >= W + 1
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= a.nlist) return;
    round_chunk<GEN>(a, a.list ? a.list[i] : i, ring_lds + threadIdx.x, nullptr);
}

// Walk kernel, few chunks (regions that cannot fall into step by themselves settle one chunk per round, and a lone lane decodes ~1 code
// per microsecond): a workgroup per chunk.  All threads first decode the residual code that WOULD start at every bit of the chunk into
// a table of lengths in LDS; thread 0 then walks the chunk and crosses every run of residuals -- three quarters of the codes -- by
// following the table.
constexpr uint32_t kCrawlThreads = 256;
template <bool GEN>
__global__ void __launch_bounds__(kCrawlThreads) derive_crawl_kernel(RoundArgs a) {
    extern __shared__ uint32_t ring_lds[];                                // ring slots (stride 64 dwords), then the table
    const uint32_t R = (uint32_t)a.W + 1u;
    uint32_t Rp = 1; while (Rp < R) Rp <<= 1;
    uint8_t* const lens = reinterpret_cast<uint8_t*>(ring_lds + (size_t)Rp * 64);
    const uint32_t c = a.list ? a.list[blockIdx.x] : blockIdx.x;
    const uint64_t lo = (uint64_t)c * kChunkBits, hi = lo + kChunkBits < a.total_bits ? lo + kChunkBits : a.total_bits;
    for (uint32_t p = threadIdx.x; p < (uint32_t)(hi - lo); p += kCrawlThreads) {
        const uint64_t pos = lo + p;
        const uint8_t* q = a.g + (pos >> 3);
        const uint64_t hi8 = __builtin_bswap64(*reinterpret_cast<const u64*>(q));
        const unsigned sh = (unsigned)pos & 7u;
        uint64_t w = hi8 << sh;
        if (sh) w |= (uint64_t)q[8] >> (8u - sh);
        uint64_t v;
        lens[p] = (uint8_t)read_code<GEN>(w, a.cod.residual, (uint32_t)a.cod.zeta_k, v);
    }
    __syncthreads();
    if (threadIdx.x == 0) round_chunk<GEN>(a, c, ring_lds, lens);
}

// Which chunks do not enter in the state their predecessor left in?  They go to the list; the first of them follows an exact chunk
// (chunk 0 is exact, and a chunk that enters in the exit state of an exact chunk is exact), so its predecessor's exit is the truth.
__global__ void derive_detect_kernel(uint32_t nchunks, uint32_t R, const WalkState* entry, const uint32_t* entry_ring, const WalkState* exit, const uint32_t* exit_ring,
                                     uint32_t* list, uint32_t* n_first) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 || c >= nchunks) return;
    bool same = same_state(entry[c], exit[c - 1]);
    const uint32_t* const er = entry_ring + (size_t)c * R; const uint32_t* const xr = exit_ring + (size_t)(c - 1) * R;
    for (uint32_t j = 0; same && j < R; j++) same = er[j] == xr[j];
    if (!same) { list[atomicAdd(&n_first[0], 1u)] = c; atomicMin(&n_first[1], c); }
}
// The mismatching chunks adopt their predecessor's exit state -- the first one always, the others unless that state is blind -- and are
// walked again; the others are dropped from the list (entry kInvalidChunk) and wait for the truth to reach them.
__global__ void derive_adopt_kernel(uint32_t R, WalkState* entry, uint32_t* entry_ring, const WalkState* exit, const uint32_t* exit_ring, uint32_t* list, const uint32_t* n_first,
                                    uint32_t* walk_list, uint32_t* n_walk) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_first[0]) return;
    const uint32_t c = list[i];
    const WalkState x = exit[c - 1];
    if (c != n_first[1] && blind_state(x)) return;
    entry[c] = x;
    for (uint32_t j = 0; j < R; j++) entry_ring[(size_t)c * R + j] = exit_ring[(size_t)(c - 1) * R + j];
    walk_list[atomicAdd(n_walk, 1u)] = c;
}

// Final pass over the exact states: record starts -> offsets[base[c] + i]; anything odd before record n is an error.
template <bool GEN>
__global__ void __launch_bounds__(64) derive_write_kernel(const uint8_t* g, uint64_t total_bits, uint32_t nchunks, int W, uint32_t minint, Codings cod,
                                                          const WalkState* entry, const uint32_t* entry_ring, const uint64_t* base, int64_t n, uint64_t* offsets, unsigned* err) {
    extern __shared__ uint32_t ring_lds[];
    const uint32_t R = (uint32_t)W + 1u;
    uint32_t M = 1; while (M < R) M <<= 1; M -= 1;
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t* const ring = ring_lds + threadIdx.x;
    WalkState st = entry[c];
    const uint32_t* const my_ring = entry_ring + (size_t)c * R;
    uint32_t cnt = M + 1;
    for (uint32_t j = 0; j < R; j++) ring[((cnt - 1 - j) & M) * 64] = my_ring[j];
    const uint64_t lo = (uint64_t)c * kChunkBits, hi = lo + kChunkBits < total_bits ? lo + kChunkBits : total_bits;
    uint64_t idx = base[c];
    unsigned bad = 0, bad_before_n = 0;
    // (a lambda cannot see `bad` change inside walk(): sample it at every record start instead)
    walk<GEN>(g, st, hi, ring, M, cnt, W, minint, cod, bad, [&](uint64_t p) {
        if (idx <= (uint64_t)n) { offsets[idx] = p; bad_before_n |= bad; }
        idx++;
    });
    if (c + 1 == nchunks && st.field == F_OUTDEG) { if (idx <= (uint64_t)n) { offsets[idx] = st.pos; bad_before_n |= bad; } idx++; }
    if (idx <= (uint64_t)n) bad_before_n |= bad;                          // the chunk ends before record n does: everything it read counts
    if (bad_before_n) atomicOr(err, ERR_MALFORMED);
    if (c + 1 == nchunks && idx <= (uint64_t)n) atomicOr(err, ERR_OVERRUN);   // fewer records than nodes
}

__global__ void derive_init_kernel(uint32_t nchunks, WalkState* entry, uint32_t* entry_ring, uint32_t R) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    WalkState st{}; st.pos = (uint64_t)c * kChunkBits; st.field = F_OUTDEG;   // the guess (chunk 0: the truth)
    entry[c] = st;
    for (uint32_t j = 0; j < R; j++) entry_ring[(size_t)c * R + j] = 0;
}

__global__ void widen_counts_kernel(const uint32_t* in, int32_t* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

}  // namespace

int derive_offsets_parallel(const uint8_t* graph, uint64_t nbytes, int64_t n, int window, int min_interval, Codings cod, uint64_t* offsets, unsigned* err,
                            hipStream_t s, int* rounds_out) {
    if (rounds_out) *rounds_out = 0;
    if (window > kMaxDeriveWindow || n <= 0 || nbytes == 0) return -1;
    const uint64_t total_bits = nbytes * 8;
    const uint64_t nch64 = (total_bits + kChunkBits - 1) / kChunkBits;
    if (nch64 > 0x7FFFFFF0ull) return -1;
    const uint32_t nchunks = (uint32_t)nch64, R = (uint32_t)window + 1u;
    const bool gen = !(cod.outdegree == BVG_GAMMA && cod.reference == BVG_UNARY && cod.block_count == BVG_GAMMA && cod.block == BVG_GAMMA && cod.residual == BVG_ZETA);
    struct Bufs {
        std::vector<void*> p;
        void* get(size_t bytes) { void* q = nullptr; if (hipMalloc(&q, bytes ? bytes : 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; } p.push_back(q); return q; }
        ~Bufs() { for (void* q : p) (void)hipFree(q); }
    } B;
    WalkState* entry = (WalkState*)B.get((size_t)nchunks * sizeof(WalkState));
    WalkState* exitst = (WalkState*)B.get((size_t)nchunks * sizeof(WalkState));
    uint32_t* entry_ring = (uint32_t*)B.get((size_t)nchunks * R * 4);
    uint32_t* exit_ring = (uint32_t*)B.get((size_t)nchunks * R * 4);
    uint32_t* counts = (uint32_t*)B.get((size_t)nchunks * 4);
    uint32_t* lists = (uint32_t*)B.get((size_t)nchunks * 2 * 4);
    uint32_t* d_n = (uint32_t*)B.get(16);                               // [0] mismatching chunks, [1] the first of them, [2] chunks to walk
    int32_t* counts_i = (int32_t*)B.get((size_t)nchunks * 4);
    uint64_t* base = (uint64_t*)B.get(((size_t)nchunks + 1) * 8);
    uint64_t* tmp = (uint64_t*)B.get(scan_tmp_elems(nchunks) * 8);
    if (!entry || !exitst || !entry_ring || !exit_ring || !counts || !lists || !d_n || !counts_i || !base || !tmp) return -2;
    hipLaunchKernelGGL(derive_init_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, s, nchunks, entry, entry_ring, R);
    uint32_t Rp = 1; while (Rp < R) Rp <<= 1;
    const size_t lds = (size_t)Rp * 64 * 4;
    const uint32_t minint = (uint32_t)min_interval;
    uint32_t* const mis_list = lists; uint32_t* const walk_list = lists + nchunks;
    auto launch_walk = [&](const uint32_t* list, uint32_t nlist, bool first) {
        RoundArgs ra{graph, total_bits, nchunks, window, minint, cod, list, nlist, entry, entry_ring, exitst, exit_ring, counts, first ? 1u : 0u};
        if (!first && nlist <= kCrawlBelow) {
            const size_t lds2 = lds + kChunkBits + 64;
            if (gen) hipLaunchKernelGGL(derive_crawl_kernel<true>, dim3(nlist), dim3(kCrawlThreads), lds2, s, ra);
            else hipLaunchKernelGGL(derive_crawl_kernel<false>, dim3(nlist), dim3(kCrawlThreads), lds2, s, ra);
        } else {
            const dim3 grid((nlist + 63) / 64), block(64);
            if (gen) hipLaunchKernelGGL(derive_round_kernel<true>, grid, block, lds, s, ra);
            else hipLaunchKernelGGL(derive_round_kernel<false>, grid, block, lds, s, ra);
        }
    };
    launch_walk(nullptr, nchunks, true);                                 // round 0: every chunk from its guess
    int rounds = 1;
    const int max_rounds = 1 << 22;
    // Rounds are bounded by PROGRESS and by TIME, not only by count: a region that settles one 4 KiB chunk per round would make the
    // rounds O(chunks) -- each with a detection pass over all chunks and two host syncs, quadratic in the stream -- before the caller
    // falls back to the sequential walk anyway.  Give up (-3) when the first mismatching chunk advanced by fewer than kSlowStep chunks
    // per round over the last kSlowRounds rounds, or when the rounds have used their time budget (30 s + 20 s per GiB of stream).
    // ONE chunk per round is not a stall: it is the crawl through the inside of a record with millions of residuals (a hub of a transposed or
    // social graph), where only the chunk behind an exact one can be settled -- a one-workgroup launch per round, seconds for a record of
    // thousands of chunks, where the sequential walk from bit 0 takes minutes per GiB (round 4 gave up there: ADVICE r4).
    constexpr uint32_t kSlowRounds = 256, kSlowStep = 1;
    uint32_t mark_round = 0, mark_first = 0;
    const auto t_start = std::chrono::steady_clock::now();
    const double budget_s = 30.0 + 20.0 * (double)nbytes / (double)(1ull << 30);
    for (;;) {
        const uint32_t init[4] = {0u, 0xFFFFFFFFu, 0u, 0u};
        if (hipMemcpyAsync(d_n, init, 16, hipMemcpyHostToDevice, s) != hipSuccess) return -2;
        hipLaunchKernelGGL(derive_detect_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, s, nchunks, R, entry, entry_ring, exitst, exit_ring, mis_list, d_n);
        uint32_t h[4] = {0, 0, 0, 0};
        if (hipMemcpyAsync(h, d_n, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -2;
        if (h[0] == 0) break;                                            // every chunk enters where its predecessor left: the one true walk
        if (rounds >= max_rounds) return -3;
        if ((uint32_t)rounds - mark_round >= kSlowRounds) {
            if (h[1] - mark_first < kSlowRounds * kSlowStep && rounds > (int)kSlowRounds) {
                if (dbg_on()) fprintf(stderr, "[bvg] derive: %u rounds moved the first unsettled chunk from %u to %u of %u: giving up on the parallel walk\n", kSlowRounds, mark_first, h[1], nchunks);
                return -3;
            }
            mark_round = (uint32_t)rounds; mark_first = h[1];
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > budget_s) return -3;
        }
        hipLaunchKernelGGL(derive_adopt_kernel, dim3((h[0] + 255) / 256), dim3(256), 0, s, R, entry, entry_ring, exitst, exit_ring, mis_list, d_n, walk_list, d_n + 2);
        uint32_t nw = 0;
        if (hipMemcpyAsync(&nw, d_n + 2, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return -2;
        if (dbg_on() && (rounds < 24 || (rounds & (rounds - 1)) == 0)) fprintf(stderr, "[bvg] derive round %d: %u of %u chunks do not enter where their predecessor left (the first: %u), %u walked again\n", rounds, h[0], nchunks, h[1], nw);
        if (nw == 0) return -3;                                          // (cannot happen: the first mismatching chunk is always walked)
        launch_walk(walk_list, nw, false);
        rounds++;
    }
    if (rounds_out) *rounds_out = rounds;
    // number the record starts and write the offsets
    hipLaunchKernelGGL(widen_counts_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, s, counts, counts_i, nchunks);
    launch_exclusive_scan(counts_i, base, nchunks, tmp, s);
    const dim3 grid((nchunks + 63) / 64), block(64);
    if (gen) hipLaunchKernelGGL(derive_write_kernel<true>, grid, block, lds, s, graph, total_bits, nchunks, window, minint, cod, entry, entry_ring, base, n, offsets, err);
    else hipLaunchKernelGGL(derive_write_kernel<false>, grid, block, lds, s, graph, total_bits, nchunks, window, minint, cod, entry, entry_ring, base, n, offsets, err);
    if (hipStreamSynchronize(s) != hipSuccess) return -2;
    return 0;
}

}  // namespace bvg
