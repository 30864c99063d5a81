// bvg_derive.hip — the offsets index from a bare .graph, in parallel (BVGraph -O / writeOffsets, BVGraph.java:2595-2609;
// loadSequential / loadOffline, BVG:1345-1464).
//
// A record's length is only known by parsing it, and parsing it needs the outdegree of the node it references (BVG:1030), so the
// reference walks the stream once, sequentially.  Here the stream is cut into CHUNKS of equal bit length and every chunk is
// walked by its own lane, starting from a GUESSED state -- "a record starts at my first bit, the window before it is empty".
// The walks are then repeated, each chunk taking as its entry state what the chunk before it reported as its exit state (first
// record start at or behind the chunk's end, the outdegrees of the last W nodes, the records seen), until no exit state changes
// any more.  Chunk 0 starts from the true state, so after round r the chunks 0..r are exact by induction: the iteration ends at
// the one true walk after at most #chunks rounds -- and in practice after a few, because a walk that starts inside a record falls
// into step with the true record boundaries within its own chunk (the codes are instantaneous and records are short).  A last pass
// numbers the records (prefix sum of the per-chunk counts) and writes the offsets.  Every step is checked: the records must add
// up to `nodes` and the last one must end in the last byte of the file, otherwise the stream is reported malformed.
#include "bvg_kernels.h"

#include <cstdio>
#include <cstdlib>

namespace bvg {

namespace {

constexpr uint64_t kMaxRecord = 1ull << 26;   // bits; a longer record (a node with millions of successors) sends the file to the sequential walk
constexpr int kDW = kMaxWindow;            // outdegree ring per lane (window sizes above it take the sequential kernel)

struct ChunkState {                        // exit state of a chunk = entry state of the next one
    uint64_t pos;                          // first record start at or behind the end of the chunk
    uint32_t records;                      // records that start inside the chunk
    uint32_t err;                          // the walk hit an impossible record (only meaningful once the states are exact)
};

// One record at cur.pos (BVG:1003-1064), field by field; returns false on an impossible record.  ring[(idx - r) & 63] = outdegree
// of the node r places back.
template <bool GEN>
__device__ __forceinline__ bool skip_record(BitCursor& cur, uint32_t* ring, uint32_t& idx, int W, int min_interval, const Codings& cod, uint64_t limit, unsigned& err) {
    const uint64_t guard = limit;
    uint64_t d = GEN ? cur.read_coded(cod.outdegree, 0, guard) : cur.read_gamma(guard);
    if (d > 0x7FFFFFFFull || cur.pos > limit) return false;
    ring[idx & 63u] = (uint32_t)d;
    if (d) {
        uint64_t ref = 0;
        if (W > 0) {
            ref = GEN ? cur.read_coded(cod.reference, 0, guard) : cur.read_unary(guard);
            if (ref > (uint64_t)W) { err |= ERR_REF_RANGE; return false; }
        }
        int64_t extra = (int64_t)d;
        if (ref > 0) {
            const uint64_t bc = GEN ? cur.read_coded(cod.block_count, 0, guard) : cur.read_gamma(guard);
            if (bc > limit - (cur.pos < limit ? cur.pos : limit) + 1) return false;
            int64_t copied = 0, tot = 0;
            for (uint64_t i = 0; i < bc; i++) {
                const int64_t b = (int64_t)(GEN ? cur.read_coded(cod.block, 0, guard) : cur.read_gamma(guard)) + (i ? 1 : 0);
                tot += b; if (!(i & 1)) copied += b;
                if (cur.pos > limit) return false;
            }
            if (!(bc & 1)) copied += (int64_t)ring[(idx - (uint32_t)ref) & 63u] - tot;      // BVG:1030
            extra = (int64_t)d - copied;
        }
        if (extra > 0 && min_interval != 0) {
            const uint64_t ic = cur.read_gamma(guard);
            if (ic > (limit - (cur.pos < limit ? cur.pos : limit)) / 2 + 1) return false;
            for (uint64_t i = 0; i < ic; i++) {
                (void)cur.read_gamma(guard);
                extra -= (int64_t)cur.read_gamma(guard) + min_interval;
                if (cur.pos > limit) return false;
            }
        }
        for (int64_t i = 0; i < extra; i++) {
            (void)(GEN ? cur.read_coded(cod.residual, (unsigned)cod.zeta_k, guard) : cur.read_zeta((unsigned)cod.zeta_k, guard));
            if (cur.pos > limit) return false;
        }
    }
    idx++;
    return cur.pos <= limit;
}

// One round: every chunk is walked from the exit state of the chunk before it (as of the previous round; in-place, so a lane may
// also see a newer one: the fixed point is the same).  offsets != nullptr: the final pass, which also writes the record starts.
template <bool GEN>
__global__ void derive_round_kernel(const uint8_t* graph, uint64_t limit_byte, uint64_t nbits, uint64_t chunk_bits, uint32_t nchunks, int W, int min_interval,
                                    Codings cod, ChunkState* st, uint32_t* win, uint32_t* changed, const uint64_t* node_base, int64_t n, uint64_t* offsets, unsigned* errp) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t ring[64];
#pragma unroll
    for (int i = 0; i < 64; i++) ring[i] = 0;
    uint64_t pos = 0; uint32_t idx = 64;
    if (c > 0) {
        pos = st[c - 1].pos;
        for (int i = 0; i < W; i++) ring[(idx - 1 - (uint32_t)i) & 63u] = win[(size_t)(c - 1) * kDW + i];     // win[.][i] = outdegree i + 1 places back
    }
    const uint64_t end = (uint64_t)(c + 1) * chunk_bits < nbits ? (uint64_t)(c + 1) * chunk_bits : nbits;
    BitCursor cur{graph, pos, limit_byte};
    uint32_t records = 0; unsigned err = 0; bool bad = false;
    uint64_t nid = offsets ? node_base[c] : 0;
    while (cur.pos < end && (!offsets || (int64_t)nid < n)) {                // (the final pass stops at the last node: what follows is padding)
        const uint64_t start = cur.pos;
        // a walk that started inside a record may read a huge count out of noise: no record is followed further than kMaxRecord bits
        const uint64_t lim = start + kMaxRecord < nbits ? start + kMaxRecord : nbits;
        if (!skip_record<GEN>(cur, ring, idx, W, min_interval, cod, lim, err)) {
            if (offsets) { bad = true; cur.pos = start; break; }              // the exact walk: a malformed stream
            cur.pos = start + 1; err = 0;                                     // a guessed walk: not a record start -- try the next bit until the walk falls into step
            continue;
        }
        if (offsets && (int64_t)nid < n) offsets[nid] = start;
        nid++; records++;
    }
    ChunkState ns{bad ? end : cur.pos, records, bad ? (err | ERR_MALFORMED) : 0u};
    bool ch = false;
    if (!offsets) {
        const ChunkState os = st[c];
        ch = os.pos != ns.pos || os.records != ns.records || os.err != ns.err;
        for (int i = 0; i < W; i++) {
            const uint32_t v = ring[(idx - 1 - (uint32_t)i) & 63u];
            if (win[(size_t)c * kDW + i] != v) { win[(size_t)c * kDW + i] = v; ch = true; }
        }
        st[c] = ns;
        if (ch) atomicAdd(changed, 1u);
    } else {
        if (ns.err) atomicOr(errp, ns.err);
        if (c == nchunks - 1) {                                             // the end of the last record = offsets[n]
            if ((int64_t)nid == n) offsets[n] = cur.pos; else atomicOr(errp, ERR_MALFORMED);
        }
    }
}

__global__ void derive_init_kernel(ChunkState* st, uint32_t* win, uint64_t chunk_bits, uint32_t nchunks, uint64_t nbits) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    const uint64_t end = (uint64_t)(c + 1) * chunk_bits < nbits ? (uint64_t)(c + 1) * chunk_bits : nbits;
    st[c] = ChunkState{end, 0u, 0u};                                         // the guess: a record starts at the first bit of every chunk
    for (int i = 0; i < kDW; i++) win[(size_t)c * kDW + i] = 0;
}

__global__ void derive_scan_kernel(const ChunkState* st, uint32_t nchunks, uint64_t* node_base, uint64_t* total) {
    // records per chunk -> first node of every chunk (one workgroup: the chunk count is small next to the stream)
    __shared__ uint64_t part[256];
    const uint32_t t = threadIdx.x, per = (nchunks + 255) / 256;
    uint64_t s = 0;
    for (uint32_t i = t * per; i < (t + 1) * per && i < nchunks; i++) s += st[i].records;
    part[t] = s;
    __syncthreads();
    if (t == 0) { uint64_t run = 0; for (int i = 0; i < 256; i++) { const uint64_t v = part[i]; part[i] = run; run += v; } *total = run; }
    __syncthreads();
    uint64_t run = part[t];
    for (uint32_t i = t * per; i < (t + 1) * per && i < nchunks; i++) { node_base[i] = run; run += st[i].records; }
}

}  // namespace

// Returns 0, or a negative value when the parallel walk cannot be used (the caller falls back to the sequential kernel):
// -1 = window too large, -2 = no memory, -3 = did not settle within the round limit.  *rounds receives the rounds taken.
int derive_offsets_parallel(const uint8_t* graph, uint64_t nbytes, int64_t n, int window, int min_interval, Codings cod, uint64_t* offsets, unsigned* d_err,
                            hipStream_t s, int* rounds_out) {
    if (window > kDW || n <= 0) return -1;
    const uint64_t nbits = nbytes * 8;
    uint64_t chunk_bits = 1ull << 16;                                        // 8 KiB of stream per lane
    while (nbits / chunk_bits > (1ull << 22)) chunk_bits <<= 1;
    const uint32_t nchunks = (uint32_t)((nbits + chunk_bits - 1) / chunk_bits);
    if (nchunks < 2) return -1;                                              // nothing to gain: one walk
    ChunkState* st = nullptr; uint32_t* win = nullptr; uint32_t* changed = nullptr; uint64_t* node_base = nullptr;
    auto done = [&](int code) { for (void* p : {(void*)st, (void*)win, (void*)changed, (void*)node_base}) if (p) (void)hipFree(p); return code; };
    if (hipMalloc(&st, (size_t)nchunks * sizeof(ChunkState)) != hipSuccess || hipMalloc(&win, (size_t)nchunks * kDW * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(&changed, 16) != hipSuccess || hipMalloc(&node_base, ((size_t)nchunks + 1) * sizeof(uint64_t)) != hipSuccess) { (void)hipGetLastError(); return done(-2); }
    const bool gen = !(cod.outdegree == BVG_GAMMA && cod.reference == BVG_UNARY && cod.block_count == BVG_GAMMA && cod.block == BVG_GAMMA && cod.residual == BVG_ZETA);
    const dim3 grid((nchunks + 63) / 64), block(64);
    hipLaunchKernelGGL(derive_init_kernel, grid, block, 0, s, st, win, chunk_bits, nchunks, nbits);
    int rounds = 0; const int max_rounds = 256;
    for (;; rounds++) {
        if (rounds >= max_rounds) return done(-3);
        if (hipMemsetAsync(changed, 0, 4, s) != hipSuccess) return done(-2);
        if (gen) hipLaunchKernelGGL((derive_round_kernel<true>), grid, block, 0, s, graph, nbytes, nbits, chunk_bits, nchunks, window, min_interval, cod, st, win, changed, (const uint64_t*)nullptr, n, (uint64_t*)nullptr, d_err);
        else hipLaunchKernelGGL((derive_round_kernel<false>), grid, block, 0, s, graph, nbytes, nbits, chunk_bits, nchunks, window, min_interval, cod, st, win, changed, (const uint64_t*)nullptr, n, (uint64_t*)nullptr, d_err);
        uint32_t ch = 0;
        if (hipMemcpyAsync(&ch, changed, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return done(-2);
        if (dbg_on() && (rounds < 12 || rounds % 32 == 0)) fprintf(stderr, "[bvg] derive round %d: %u of %u chunks changed\n", rounds, ch, nchunks);
        if (!ch) break;
    }
    if (rounds_out) *rounds_out = rounds + 1;
    hipLaunchKernelGGL(derive_scan_kernel, dim3(1), dim3(256), 0, s, st, nchunks, node_base, node_base + nchunks);
    uint64_t total = 0;
    if (hipMemcpyAsync(&total, node_base + nchunks, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return done(-2);
    if (total != (uint64_t)n) {                                              // the records do not add up to `nodes`: malformed (or the wrong properties)
        const unsigned e = ERR_MALFORMED;
        (void)hipMemcpyAsync(d_err, &e, sizeof e, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s);
        return done(0);
    }
    if (gen) hipLaunchKernelGGL((derive_round_kernel<true>), grid, block, 0, s, graph, nbytes, nbits, chunk_bits, nchunks, window, min_interval, cod, st, win, changed, node_base, n, offsets, d_err);
    else hipLaunchKernelGGL((derive_round_kernel<false>), grid, block, 0, s, graph, nbytes, nbits, chunk_bits, nchunks, window, min_interval, cod, st, win, changed, node_base, n, offsets, d_err);
    if (hipStreamSynchronize(s) != hipSuccess) return done(-2);
    return done(0);
}

}  // namespace bvg
