// bvg_derive_seq.hip — the offsets index from a bare .graph by ONE wavefront walking the records in order: the fall-back of the
// chunk-parallel derivation (bvg_derive.hip) for windows > 127, codes longer than 64 bits and streams the parallel walk finds odd.
// (Until round 4 this lived beside the experimental streaming kernel in bvg_stream.hip; that kernel is now experimental/bvg_stream.hip.)
#include "bvg_kernels.h"
#include "bvg_lds_codes.h"

namespace bvg {

namespace {


// ---------------------------------------------------------------------------------------------------
// Offsets index from a bare .graph: the records have to be walked one after the other (a record's
// length is only known by parsing it: BVG:1003-1064), so ONE wavefront does it, all 64 lanes executing the
// same parse in step over an LDS ring of the stream (every LDS read is a broadcast); the lanes are
// used for the coalesced refills of the ring and for writing the offsets 64 at a time.
constexpr uint32_t kDerWords = 2048, kDerMask = kDerWords - 1;


template <bool GEN>
__global__ void __launch_bounds__(64) derive_offsets_kernel(const uint8_t* graph, uint64_t padded_bytes, uint64_t nbytes, int64_t n, int window,
                                                            int min_interval, Codings cod, uint64_t* offsets, unsigned* errp) {
    __shared__ __attribute__((aligned(16))) uint32_t ring[kDerWords];
    __shared__ uint32_t dring[kRingBig];                // outdegrees of the last nodes (BVG:1030 needs outdegree(x - ref))
    const unsigned lane = threadIdx.x;
    const uint64_t total_bits = nbytes * 8;
    uint64_t whi = 0;                         // bits [whi - kDerBits, whi) are in the ring
    uint64_t pos = 0;
    unsigned err = 0;
    const uint32_t zk = (uint32_t)cod.zeta_k;
    for (unsigned i = lane; i < (unsigned)kRingBig; i += 64) dring[i] = 0;
    __syncthreads();
    auto ensure = [&](uint64_t upto) {        // make bits [pos, upto) available (upto - pos < ring size)
        while (whi < upto) {
            const uint64_t byte = (whi >> 3) + ((uint64_t)lane << 4);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (byte + 16 <= padded_bytes) v = *reinterpret_cast<const uint4*>(graph + byte);
            uint4 w; w.x = __builtin_bswap32(v.x); w.y = __builtin_bswap32(v.y); w.z = __builtin_bswap32(v.z); w.w = __builtin_bswap32(v.w);
            *reinterpret_cast<uint4*>(&ring[((uint32_t)(whi >> 5) + (lane << 2)) & kDerMask]) = w;
            whi += 8192;
            __syncthreads();
        }
    };
    auto code = [&](int coding, uint32_t k, uint64_t& val) -> bool {   // one code at pos; advances pos
        ensure(pos + 160);
        const uint64_t w = bvg::win64<kDerMask>(ring, (uint32_t)pos);       // ring index uses the low bits only
        uint32_t len;
        if (coding == BVG_UNARY) {                                      // unary values may exceed one 64-bit window
            uint64_t z = 0, ww = w;
            while (ww == 0 && pos + 64 <= total_bits) { pos += 64; z += 64; ensure(pos + 160); ww = bvg::win64<kDerMask>(ring, (uint32_t)pos); }
            const uint32_t lz = ww ? (uint32_t)__builtin_clzll(ww) : 64u;
            val = z + lz; len = lz < 64 ? lz + 1 : 0;
        }
        else if (GEN) len = decode_generic_w(w, coding, k, &val);
        else if (coding == BVG_ZETA) len = zeta64(w, k, val);
        else len = gamma64(w, val);
        pos += len;
        return len != 0 && pos <= total_bits;
    };
    uint64_t mine = 0;                        // offset of node (x0 + lane) for the coalesced write-out
    for (int64_t x = 0; x < n && !err; x++) {
        if ((unsigned)(x & 63) == lane) mine = pos;
        uint64_t v;
        if (!code(cod.outdegree, 0, v) || v > 0x7FFFFFFFull) { err = ERR_OVERRUN; break; }
        const uint32_t d = (uint32_t)v;
        dring[(uint32_t)x & (uint32_t)(kRingBig - 1)] = d;
        if (d > 0) {
            uint32_t ref = 0;
            if (window > 0) {
                if (!code(cod.reference, 0, v)) { err = ERR_OVERRUN; break; }
                if (v > (uint64_t)window || (int64_t)v > x) { err = ERR_REF_RANGE; break; }
                ref = (uint32_t)v;
            }
            int64_t extra = d;
            if (ref > 0) {
                if (!code(cod.block_count, 0, v)) { err = ERR_OVERRUN; break; }
                const uint64_t bc = v;
                int64_t copied = 0, tot = 0;
                for (uint64_t i = 0; i < bc; i++) {
                    if (!code(cod.block, 0, v)) { err = ERR_OVERRUN; break; }
                    const int64_t b = (int64_t)v + (i ? 1 : 0);
                    tot += b; if (!(i & 1)) copied += b;
                }
                if (err) break;
                if (!(bc & 1)) copied += (int64_t)dring[(uint32_t)(x - ref) & (uint32_t)(kRingBig - 1)] - tot;
                extra = (int64_t)d - copied;
                if (extra < 0 || copied < 0) { err = ERR_MALFORMED; break; }
            }
            if (extra > 0 && min_interval != 0) {
                if (!code(BVG_GAMMA, 0, v)) { err = ERR_OVERRUN; break; }
                const uint64_t ic = v;
                for (uint64_t i = 0; i < ic; i++) {
                    uint64_t v2;
                    if (!code(BVG_GAMMA, 0, v) || !code(BVG_GAMMA, 0, v2)) { err = ERR_OVERRUN; break; }
                    extra -= (int64_t)v2 + min_interval;
                }
                if (err) break;
                if (extra < 0) { err = ERR_MALFORMED; break; }
            }
            for (int64_t i = 0; i < extra; i++) if (!code(cod.residual, zk, v)) { err = ERR_OVERRUN; break; }
            if (err) break;
        }
        if ((x & 63) == 63 || x == n - 1) {                                  // 64 offsets at a time, coalesced
            const int64_t x0 = x & ~63ll;
            if (x0 + lane <= x) offsets[x0 + lane] = mine;
        }
    }
    if (lane == 0) { offsets[n] = pos; if (err) atomicOr(errp, err); }
}

}  // namespace

void launch_derive_offsets(const uint8_t* graph, uint64_t padded_bytes, uint64_t nbytes, int64_t n, int window, int min_interval, Codings cod,
                           uint64_t* offsets, unsigned* err, hipStream_t s) {
    const bool gen = !(cod.outdegree == BVG_GAMMA && cod.reference == BVG_UNARY && cod.block_count == BVG_GAMMA &&
                       cod.block == BVG_GAMMA && cod.residual == BVG_ZETA);
    if (gen) hipLaunchKernelGGL((derive_offsets_kernel<true>), dim3(1), dim3(64), 0, s, graph, padded_bytes, nbytes, n, window, min_interval, cod, offsets, err);
    else hipLaunchKernelGGL((derive_offsets_kernel<false>), dim3(1), dim3(64), 0, s, graph, padded_bytes, nbytes, n, window, min_interval, cod, offsets, err);
}

}  // namespace bvg
