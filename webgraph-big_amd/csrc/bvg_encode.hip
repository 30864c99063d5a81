// bvg_encode.hip — BVGraph.store on the device (second half): the compressor of
// BVGraph.java:1595-1618 (intervalize), :1977-2159 (diffComp), :2216-2327 (CompressionThread.call: reference selection), from an
// adjacency in CSR form to the .graph bit stream and its offsets, byte for byte what the reference writes.
//
// The reference compresses node after node: for every node it tries every reference within the window by a dry run of diffComp
// (W + 1 trial compressions), keeps the cheapest admissible one (ties: the nearest; admissible: the referenced list is not empty
// and its reference chain is shorter than maxRefCount) and writes the record.  Only the ADMISSIBILITY depends on earlier
// decisions (the chain lengths); the trial sizes do not.  So:
//   E1  one lane per (node, reference) pair: the size in bits of diffComp(node, reference) -- a streaming two-pointer walk over
//       the two lists with an on-line intervaliser, no per-lane arrays (code lengths add up in any order);
//   E2  one lane per CHUNK (the reference's per-thread node ranges, BVG:2404-2457: a chunk starts with an empty window; one chunk =
//       the single-threaded store): the sequential choice over the size table, which is all that is sequential;
//   E3  record sizes -> bit offsets (prefix sum);
//   E4  one lane per node: diffComp once more for the chosen reference, first sizing the record's three sections (copy blocks,
//       intervals, residuals), then writing all three in one walk with three cursors; bits are OR-ed into the zeroed output with
//       32-bit atomics (records are not byte aligned, neighbours share words).
// Host side: bvg_store (include/bvgraph_hip.h) uploads the adjacency, runs E1-E4 and returns the stream and the offsets.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "bvg_kernels.h"

namespace bvg {

namespace {

struct EncParams {
    int W, max_ref, min_interval, zeta_k;
    int outdegree_coding, block_coding, residual_coding, reference_coding, block_count_coding;
    int64_t n, chunk_nodes;
};

__device__ __forceinline__ int msb64(uint64_t x) { return 63 - (int)__builtin_clzll(x); }
__device__ __forceinline__ uint64_t int2nat(int64_t v) { return v >= 0 ? (uint64_t)v << 1 : (((uint64_t)(-(v + 1))) << 1) + 1; }   // Fast.int2nat

// ---- code lengths (dsiutils OutputBitStream) ----
__device__ __forceinline__ uint64_t len_gamma(uint64_t x) { return 2u * (uint64_t)msb64(x + 1) + 1u; }
__device__ __forceinline__ uint64_t len_coded(uint64_t x, int coding, int k) {
    switch (coding) {
        case BVG_UNARY: return x + 1;
        case BVG_DELTA: { const int b = msb64(x + 1); return len_gamma((uint64_t)b) + (uint64_t)b; }
        case BVG_ZETA: { const uint64_t v = x + 1; const int h = msb64(v) / k; const uint64_t left = 1ull << (h * k); return (uint64_t)h + 1 + (v - left < left ? (uint64_t)(h * k + k - 1) : (uint64_t)(h * k + k)); }
        case BVG_NIBBLE: return x == 0 ? 4u : (uint64_t)(msb64(x) / 3 + 1) * 4u;
        case BVG_GOLOMB: { if (k == 0) return 0; const uint64_t b = (uint64_t)k; uint64_t l = x / b + 1; if (b == 1) return l; const int lg = msb64(b); const uint64_t thr = (1ull << (lg + 1)) - b; return l + ((x % b) < thr ? (uint64_t)lg : (uint64_t)lg + 1); }
        default: return len_gamma(x);
    }
}

// ---- bit sink: MSB-first, at an arbitrary bit position of a zeroed buffer, OR-ed in with 32-bit atomics ----
struct BitOut {
    uint32_t* out; uint64_t pos;
    __device__ __forceinline__ void put(uint64_t v, int n) {              // the n <= 64 low bits of v
        while (n > 0) {
            const uint64_t w = pos >> 5; const int o = (int)(pos & 31u);
            const int take = n < 32 - o ? n : 32 - o;
            const uint32_t piece = (uint32_t)((v >> (n - take)) & (take == 32 ? 0xFFFFFFFFull : ((1ull << take) - 1ull)));
            if (piece) atomicOr(&out[w], __builtin_bswap32(piece << (32 - o - take)));
            pos += (uint64_t)take; n -= take;
        }
    }
    __device__ __forceinline__ void zeros(uint64_t n) { pos += n; }
    __device__ __forceinline__ void unary(uint64_t x) { zeros(x); put(1, 1); }
    __device__ __forceinline__ void gamma(uint64_t x) { const int b = msb64(x + 1); unary((uint64_t)b); if (b) put((x + 1) & ((1ull << b) - 1), b); }
    __device__ __forceinline__ void coded(uint64_t x, int coding, int k) {
        switch (coding) {
            case BVG_UNARY: unary(x); break;
            case BVG_DELTA: { const int b = msb64(x + 1); gamma((uint64_t)b); if (b) put((x + 1) & ((1ull << b) - 1), b); break; }
            case BVG_ZETA: { const uint64_t v = x + 1; const int h = msb64(v) / k; const uint64_t left = 1ull << (h * k); unary((uint64_t)h); if (v - left < left) put(v - left, h * k + k - 1); else put(v, h * k + k); break; }
            case BVG_NIBBLE: { if (x == 0) { put(8, 4); break; } int h = msb64(x) / 3; do { put(h == 0, 1); put((x >> (h * 3)) & 7, 3); } while (h-- != 0); break; }
            case BVG_GOLOMB: { if (k == 0) break; const uint64_t b = (uint64_t)k; unary(x / b); if (b == 1) break; const int l = msb64(b); const uint64_t thr = (1ull << (l + 1)) - b, r = x % b; if (r < thr) put(r, l); else put(r + thr, l + 1); break; }
            default: gamma(x);
        }
    }
};

// The walk of diffComp (BVG:1996-2051) with the intervaliser (BVG:1595-1618) on line: calls c.block(len) for every copy block in order,
// c.interval(left, len) / c.residual(v) for the extras in increasing order (each of the two sequences in its own order).
template <class C>
__device__ __forceinline__ void diff_walk(const EncParams& p, int ref, const int64_t* rl, int64_t rlen, const int64_t* cl, int64_t clen, C& c) {
    int64_t j = 0, k = 0, blk = 0; bool copying = true;
    if (ref == 0) rlen = 0;
    int64_t run_s = 0, run_l = 0;                                          // the current run of consecutive extras
    auto flush = [&]() {
        if (run_l == 0) return;
        if (p.min_interval != 0 && run_l >= p.min_interval) c.interval(run_s, run_l);
        else for (int64_t t = 0; t < run_l; t++) c.residual(run_s + t);
        run_l = 0;
    };
    auto extra = [&](int64_t v) { if (run_l && v == run_s + run_l) run_l++; else { flush(); run_s = v; run_l = 1; } };
    while (j < clen && k < rlen) {
        const int64_t a = cl[j], b = rl[k];
        if (copying) {
            if (a > b) { c.block(blk); copying = false; blk = 0; }
            else if (a < b) { extra(a); j++; }
            else { j++; k++; blk++; }
        } else {
            if (a < b) { extra(a); j++; }
            else if (a > b) { k++; blk++; }
            else { c.block(blk); copying = true; blk = 0; }
        }
    }
    if (copying && k < rlen) c.block(blk);
    while (j < clen) extra(cl[j++]);
    flush();
}

struct SizeAcc {                                                           // the dry run: section sizes in bits
    const EncParams& p; int64_t x;
    uint64_t nblocks = 0, bits_blocks = 0, ic = 0, bits_iv = 0, nres = 0, bits_res = 0, nextra = 0; int64_t prev_iv = 0, prev_res = 0;
    __device__ SizeAcc(const EncParams& pp, int64_t xx) : p(pp), x(xx) {}
    __device__ __forceinline__ void block(int64_t b) { bits_blocks += len_coded((uint64_t)(nblocks ? b - 1 : b), p.block_coding, 0); nblocks++; }
    __device__ __forceinline__ void interval(int64_t l, int64_t n) {
        bits_iv += len_gamma(ic ? (uint64_t)(l - prev_iv - 1) : int2nat(l - x)) + len_gamma((uint64_t)(n - p.min_interval));
        prev_iv = l + n; ic++; nextra += (uint64_t)n;
    }
    __device__ __forceinline__ void residual(int64_t r) {
        bits_res += len_coded(nres ? (uint64_t)(r - prev_res - 1) : int2nat(r - x), p.residual_coding, p.zeta_k);
        prev_res = r; nres++; nextra++;
    }
    __device__ __forceinline__ uint64_t head_bits(int ref) const {        // reference + block count
        return (p.W > 0 ? len_coded((uint64_t)ref, p.reference_coding, 0) : 0u) + (ref ? len_coded(nblocks, p.block_count_coding, 0) : 0u);
    }
    __device__ __forceinline__ uint64_t ic_bits() const { return (nextra && p.min_interval != 0) ? len_gamma(ic) : 0u; }
    __device__ __forceinline__ uint64_t total(int ref) const { return head_bits(ref) + (ref ? bits_blocks : 0u) + ic_bits() + bits_iv + bits_res; }
};

struct WriteAcc {                                                          // the real run: three cursors
    const EncParams& p; int64_t x; BitOut wb, wi, wr;
    uint64_t nblocks = 0, ic = 0, nres = 0; int64_t prev_iv = 0, prev_res = 0;
    __device__ WriteAcc(const EncParams& pp, int64_t xx, uint32_t* out, uint64_t pb, uint64_t pi, uint64_t pr) : p(pp), x(xx), wb{out, pb}, wi{out, pi}, wr{out, pr} {}
    __device__ __forceinline__ void block(int64_t b) { wb.coded((uint64_t)(nblocks ? b - 1 : b), p.block_coding, 0); nblocks++; }
    __device__ __forceinline__ void interval(int64_t l, int64_t n) {
        wi.gamma(ic ? (uint64_t)(l - prev_iv - 1) : int2nat(l - x)); wi.gamma((uint64_t)(n - p.min_interval));
        prev_iv = l + n; ic++;
    }
    __device__ __forceinline__ void residual(int64_t r) {
        wr.coded(nres ? (uint64_t)(r - prev_res - 1) : int2nat(r - x), p.residual_coding, p.zeta_k);
        prev_res = r; nres++;
    }
};

__device__ __forceinline__ int64_t chunk_first(const EncParams& p, int64_t x) { return p.chunk_nodes > 0 ? (x / p.chunk_nodes) * p.chunk_nodes : 0; }

// E1: size table.  sizes[x * (W+1) + r] = bits of diffComp(x, ref = r), 0xFFFFFFFF where r is impossible whatever the chains.
__global__ void enc_sizes_kernel(EncParams p, const uint64_t* adj_off, const int64_t* adj, uint32_t* sizes) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cyc = p.W + 1;
    const int64_t x = idx / cyc; const int r = (int)(idx % cyc);
    if (x >= p.n) return;
    const int64_t d = (int64_t)(adj_off[x + 1] - adj_off[x]);
    uint32_t out = 0xFFFFFFFFu;
    if (d > 0 && (r == 0 || (x - r >= chunk_first(p, x) && adj_off[x - r + 1] > adj_off[x - r]))) {
        SizeAcc acc(p, x);
        const int64_t y = x - r;
        diff_walk(p, r, adj + adj_off[y], (int64_t)(adj_off[y + 1] - adj_off[y]), adj + adj_off[x], d, acc);
        const uint64_t t = acc.total(r);
        out = t < 0xFFFFFFFFull ? (uint32_t)t : 0xFFFFFFFEu;
    }
    sizes[idx] = out;
}

// the chunk's one wavefront orders its own LDS traffic (its lanes run in lock step; only the compiler has to be held back)
__device__ __forceinline__ void enc_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// E2: the sequential choice (BVG:2254-2270), one WAVEFRONT per chunk.  The choice for node x needs the chain lengths of the W nodes
// before it, so a chunk is a serial walk -- but not a walk against global memory: the size table of 64 nodes at a time is staged in
// LDS with coalesced loads, lane r weighs reference r (admissible? its size), a wave minimum over {size, r} picks the cheapest with
// the nearest winning ties, lane 0 updates the chain-length ring, and the 64 results leave in one store.  ~0.1 us per node instead of
// the ~2 us of one lane chasing the table through HBM (a single-range store of 1 M nodes: 2.2 s -> 0.1 s).
// best[x] = chosen reference; recbits[x] = bits of the whole record.
__global__ void __launch_bounds__(64) enc_choose_kernel(EncParams p, const uint64_t* adj_off, const uint32_t* sizes, uint8_t* best, int32_t* recbits) {
    extern __shared__ uint32_t enc_lds[];                                  // tile[64][cyc] | refc[cyc] | d[64] | best[64] | bits[64]
    const int cyc = p.W + 1;
    uint32_t* const tile = enc_lds;
    int32_t* const refc = reinterpret_cast<int32_t*>(enc_lds + 64 * cyc);
    uint32_t* const dl = enc_lds + 64 * cyc + cyc;
    uint32_t* const bl = dl + 64;
    uint32_t* const rb = bl + 64;
    const unsigned lane = threadIdx.x;
    const int64_t c = blockIdx.x;
    const int64_t cn = p.chunk_nodes > 0 ? p.chunk_nodes : p.n;
    const int64_t first = c * cn;
    if (first >= p.n) return;
    const int64_t last = first + cn < p.n ? first + cn : p.n;
    for (int i = (int)lane; i < cyc; i += 64) refc[i] = 0;
    const int64_t max_ref = p.max_ref < 0 ? (int64_t)0x7FFFFFFF : p.max_ref;
    for (int64_t x0 = first; x0 < last; x0 += 64) {
        const int cntn = (int)(last - x0 < 64 ? last - x0 : 64);
        __syncthreads();
        for (int t = (int)lane; t < cntn * cyc; t += 64) tile[t] = sizes[x0 * cyc + t];
        if ((int)lane < cntn) dl[lane] = (uint32_t)(adj_off[x0 + lane + 1] - adj_off[x0 + lane]);
        __syncthreads();
        for (int i = 0; i < cntn; i++) {
            const int64_t x = x0 + i;
            const uint32_t d = dl[i];
            const int ci = (int)((x - first) % cyc);
            uint32_t bits = (uint32_t)len_coded((uint64_t)d, p.outdegree_coding, 0);
            uint32_t b = 0;
            if (d > 0) {
                if (lane == 0) refc[ci] = -1;                              // (the list itself: "no reference" is always admissible)
                enc_wave_sync();
                // lanes weigh 64 references at a time: the smallest admissible size by a wave minimum (0 - max of the complement), the
                // nearest reference among its holders by a ballot; further groups of 64 (windows above 63) only win with a smaller size
                uint32_t bsz = 0xFFFFFFFFu; int br = 0, bcand = ci;
                for (int r0 = 0; r0 < cyc; r0 += 64) {
                    const int r = r0 + (int)lane;
                    uint32_t sz = 0xFFFFFFFFu; int cand = 0;
                    if (r < cyc && !(x - r < first && r != 0)) {
                        cand = (int)(((x - first) - r + 2ll * cyc) % cyc);
                        const uint32_t t = tile[i * cyc + r];
                        if (refc[cand] < max_ref) sz = t;
                    }
                    const uint32_t m = ~wave_max32(~sz);
                    if (m < bsz) {
                        const int wl = __ffsll((unsigned long long)ballot(sz == m)) - 1;
                        bsz = m; br = r0 + wl; bcand = __shfl(cand, wl, 64);
                    }
                }
                b = (uint32_t)br;
                bits += bsz;
                enc_wave_sync();
                if (lane == 0) refc[ci] = refc[bcand] + 1;
            } else if (lane == 0) refc[ci] = 0;                            // (an empty list is never referenced; its slot just leaves the window)
            if (lane == 0) { bl[i] = b; rb[i] = bits; }
            enc_wave_sync();
        }
        if ((int)lane < cntn) { best[x0 + lane] = (uint8_t)bl[lane]; recbits[x0 + lane] = (int32_t)rb[lane]; }
    }
}

// E4: the records.
__global__ void enc_write_kernel(EncParams p, const uint64_t* adj_off, const int64_t* adj, const uint8_t* best, const uint64_t* offsets, uint32_t* out) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= p.n) return;
    const int64_t d = (int64_t)(adj_off[x + 1] - adj_off[x]);
    BitOut w{out, offsets[x]};
    w.coded((uint64_t)d, p.outdegree_coding, 0);
    if (d == 0) return;
    const int r = best[x];
    const int64_t y = x - r;
    const int64_t* rl = adj + adj_off[y]; const int64_t rlen = (int64_t)(adj_off[y + 1] - adj_off[y]);
    const int64_t* cl = adj + adj_off[x];
    SizeAcc acc(p, x);
    diff_walk(p, r, rl, rlen, cl, d, acc);
    if (p.W > 0) w.coded((uint64_t)r, p.reference_coding, 0);
    if (r) w.coded(acc.nblocks, p.block_count_coding, 0);
    const uint64_t pb = w.pos, pic = pb + (r ? acc.bits_blocks : 0u);
    BitOut wic{out, pic};
    if (acc.nextra && p.min_interval != 0) wic.gamma(acc.ic);
    const uint64_t pi = wic.pos, pr = pi + acc.bits_iv;
    WriteAcc wa(p, x, out, pb, pi, pr);
    diff_walk(p, r, rl, rlen, cl, d, wa);
}

__global__ void enc_check_kernel(const uint64_t* adj_off, const int64_t* adj, int64_t n, unsigned* bad) {
    // successor lists must be strictly increasing and inside [0, n) (the reference throws on a duplicate, BVG:2141)
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const uint64_t a = adj_off[x], b = adj_off[x + 1];
    if (b < a || b - a > 0x7FFFFFFFull) { atomicOr(bad, 1u); return; }
    for (uint64_t i = a; i < b; i++) { const int64_t v = adj[i]; if (v < 0 || v >= n || (i > a && adj[i - 1] >= v)) { atomicOr(bad, 2u); return; } }
}

}  // namespace

// Device-side store: adjacency already in HBM.  d_graph_out / d_offsets_out are hipMalloc'ed here (caller frees with hipFree);
// *graph_bytes = ceil(offsets[n] / 8).  Returns a bvg status.
int encode_store_dev(const bvg_params& bp, const uint64_t* d_adj_off, const int64_t* d_adj, int64_t n, int64_t chunk_nodes, hipStream_t s,
                     uint8_t** d_graph_out, uint64_t* graph_bytes, uint64_t** d_offsets_out) {
    if (bp.window_size > 127) return BVG_E_UNSUPPORTED;
    EncParams p{bp.window_size, bp.max_ref_count, bp.min_interval_length, bp.zeta_k, bp.outdegree_coding, bp.block_coding, bp.residual_coding,
                bp.reference_coding, bp.block_count_coding, n, chunk_nodes > 0 ? chunk_nodes : 0};
    const int cyc = p.W + 1;
    uint32_t* sizes = nullptr; uint8_t* best = nullptr; int32_t* recbits = nullptr; uint64_t* offsets = nullptr; uint64_t* tmp = nullptr; unsigned* bad = nullptr;
    uint8_t* graph = nullptr;
    auto done = [&](int code) {
        for (void* q : {(void*)sizes, (void*)best, (void*)recbits, (void*)tmp, (void*)bad}) if (q) (void)hipFree(q);
        if (code) { if (offsets) (void)hipFree(offsets); if (graph) (void)hipFree(graph); }
        return code;
    };
    const size_t nn = (size_t)(n > 0 ? n : 1);
#define ENC_CHK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { (void)hipGetLastError(); return done(_e == hipErrorOutOfMemory ? BVG_E_NOMEM : BVG_E_HIP); } } while (0)
    ENC_CHK(hipMalloc(&sizes, nn * (size_t)cyc * sizeof(uint32_t)));
    ENC_CHK(hipMalloc(&best, nn));
    ENC_CHK(hipMalloc(&recbits, nn * sizeof(int32_t)));
    ENC_CHK(hipMalloc(&offsets, (nn + 1) * sizeof(uint64_t)));
    ENC_CHK(hipMalloc(&tmp, scan_tmp_elems((int64_t)nn) * sizeof(uint64_t)));
    ENC_CHK(hipMalloc(&bad, sizeof(unsigned)));
    ENC_CHK(hipMemsetAsync(bad, 0, sizeof(unsigned), s));
    ENC_CHK(hipMemsetAsync(offsets, 0, (nn + 1) * sizeof(uint64_t), s));
    uint64_t total_bits = 0;
    if (n > 0) {
        hipLaunchKernelGGL(enc_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_adj_off, d_adj, n, bad);
        unsigned hb = 0;
        ENC_CHK(hipMemcpyAsync(&hb, bad, sizeof hb, hipMemcpyDeviceToHost, s));
        ENC_CHK(hipStreamSynchronize(s));
        if (hb) return done(BVG_E_ARG);
        const int64_t pairs = n * cyc;
        hipLaunchKernelGGL(enc_sizes_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, p, d_adj_off, d_adj, sizes);
        const int64_t cn = p.chunk_nodes > 0 ? p.chunk_nodes : n;
        const int64_t nchunks = (n + cn - 1) / cn;
        hipLaunchKernelGGL(enc_choose_kernel, dim3((unsigned)nchunks), dim3(64), (size_t)(64 * cyc + cyc + 3 * 64) * sizeof(uint32_t), s, p, d_adj_off, sizes, best, recbits);
        launch_exclusive_scan(recbits, offsets, n, tmp, s);
        ENC_CHK(hipMemcpyAsync(&total_bits, offsets + n, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
        ENC_CHK(hipStreamSynchronize(s));
    }
    const uint64_t nbytes = (total_bits + 7) / 8;
    const size_t alloc = (size_t)((nbytes + 15) & ~15ull) + 64;           // zero padded: the decoder's loads may run past the end
    ENC_CHK(hipMalloc(&graph, alloc));
    ENC_CHK(hipMemsetAsync(graph, 0, alloc, s));
    if (n > 0) hipLaunchKernelGGL(enc_write_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, d_adj_off, d_adj, best, offsets, reinterpret_cast<uint32_t*>(graph));
    ENC_CHK(hipStreamSynchronize(s));
#undef ENC_CHK
    *d_graph_out = graph; *graph_bytes = nbytes; *d_offsets_out = offsets;
    return done(0);
}

}  // namespace bvg
