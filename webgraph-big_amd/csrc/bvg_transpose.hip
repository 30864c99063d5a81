// bvg_transpose.hip — transposition feed.
//
// Transform.transposeOffline (Transform.java:1058-1160) scans the graph, collects (target, source) pairs in batches, sorts every
// batch on the CPU (processBatch, Transform.java:938) and merges the batches.  Here the scan is the HIP decode (successors stay
// in HBM), the pairs are sorted by target on the device with a stable LSD radix sort (rocPRIM device_radix_sort — a plain
// library sort, like the reference's use of fastutil's sort), and the result is the transpose in CSR form: for every node y
// the sources x of its incoming arcs in increasing order.  Arcs are produced in source-major order, so a stable sort on the
// target alone leaves every list sorted.  Only the low ceil(log2 n) key bits are sorted.
#include <cstdint>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "bvg_kernels.h"

namespace bvg {

namespace {

// one wavefront per 64 consecutive nodes: the lanes write node x's id over its successor range (coalesced) and check that
// every target is a node of the graph (the sort below only looks at the low ceil(log2 n) bits of a target)
__global__ void expand_sources_kernel(const uint64_t* cum, int64_t n, const int64_t* succ, int64_t* src, unsigned* bad) {
    const int64_t x0 = (int64_t)blockIdx.x * 64;
    const unsigned lane = threadIdx.x;
    const int64_t xe = x0 + 64 < n ? x0 + 64 : n;
    bool oob = false;
    const uint64_t lo = cum[x0], hi = cum[xe];
    // walk the arcs of the 64 nodes in chunks of 64; the owner of arc t = number of node ends <= t (ends are non-decreasing)
    __shared__ uint64_t ends[64];
    ends[lane] = x0 + (int64_t)lane < xe ? cum[x0 + lane + 1] : ~0ull;
    __syncthreads();
    const int cnt = (int)(xe - x0);
    for (uint64_t t = lo + lane; t < hi; t += 64) {
        int l = 0, r = cnt;
        while (l < r) { const int m = (l + r) >> 1; if (ends[m] <= t) l = m + 1; else r = m; }
        src[t] = x0 + l;
        const int64_t y = succ[t];
        oob |= y < 0 || y >= n;
    }
    if (oob) atomicOr(bad, 1u);
}

// toffsets[y] = number of arcs with target < y = lower bound of y in the sorted targets
__global__ void offsets_from_sorted_kernel(const uint64_t* keys, uint64_t m, int64_t n, uint64_t* toffsets) {
    const int64_t y = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (y > n) return;
    uint64_t l = 0, r = m;
    while (l < r) { const uint64_t mid = (l + r) >> 1; if (keys[mid] < (uint64_t)y) l = mid + 1; else r = mid; }
    toffsets[y] = l;
}

// Transform.union (Transform.java: the union of a graph and its transpose is symmetrizeOffline, :573-575): per node the sorted
// union of two increasing lists, equal elements once.  One thread per node; COUNT pass sizes the lists, WRITE pass fills them.
template <bool WRITE>
__global__ void __launch_bounds__(256) union_lists_kernel(const uint64_t* acum, const int64_t* asucc, const uint64_t* bcum, const int64_t* bsucc, int64_t n,
                                                          int32_t* cnt, const uint64_t* ocum, int64_t* out) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    uint64_t i = acum[x], j = bcum[x]; const uint64_t ie = acum[x + 1], je = bcum[x + 1];
    uint64_t k = WRITE ? ocum[x] : 0;
    while (i < ie || j < je) {
        const int64_t a = i < ie ? asucc[i] : INT64_MAX, b = j < je ? bsucc[j] : INT64_MAX;
        const int64_t m = a < b ? a : b;
        if (WRITE) out[k] = m;
        k++;
        i += a == m; j += b == m;
    }
    if (!WRITE) cnt[x] = (int32_t)k;
}

}  // namespace

void launch_union_count(const uint64_t* acum, const int64_t* asucc, const uint64_t* bcum, const int64_t* bsucc, int64_t n, int32_t* cnt, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL((union_lists_kernel<false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acum, asucc, bcum, bsucc, n, cnt, (const uint64_t*)nullptr, (int64_t*)nullptr);
}
void launch_union_write(const uint64_t* acum, const int64_t* asucc, const uint64_t* bcum, const int64_t* bsucc, int64_t n, const uint64_t* ocum, int64_t* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL((union_lists_kernel<true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acum, asucc, bcum, bsucc, n, (int32_t*)nullptr, ocum, out);
}

size_t transpose_temp_bytes(uint64_t arcs, int64_t n) {
    size_t sort_b = 0;
    const unsigned bits = n > 1 ? 64u - (unsigned)__builtin_clzll((unsigned long long)(n - 1)) : 1u;
    (void)rocprim::radix_sort_pairs(nullptr, sort_b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int64_t*)nullptr, (int64_t*)nullptr, (size_t)arcs, 0u, bits, (hipStream_t)0);
    return sort_b;
}

// succ[arcs] (targets, source-major), cum[n+1] -> toffsets[n+1], tsucc[arcs]; work buffers: src[arcs], keys_out[arcs], temp
hipError_t transpose_pairs(const uint64_t* cum, int64_t n, uint64_t arcs, const int64_t* succ, int64_t* src, uint64_t* keys_out, void* temp, size_t temp_bytes,
                           uint64_t* toffsets, int64_t* tsucc, unsigned* d_bad, hipStream_t s) {
    if (arcs == 0) return hipMemsetAsync(toffsets, 0, (size_t)(n + 1) * sizeof(uint64_t), s);
    hipLaunchKernelGGL(expand_sources_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, cum, n, succ, src, d_bad);
    const unsigned bits = n > 1 ? 64u - (unsigned)__builtin_clzll((unsigned long long)(n - 1)) : 1u;
    size_t tb = temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(temp, tb, reinterpret_cast<const uint64_t*>(succ), keys_out, src, tsucc, (size_t)arcs, 0u, bits, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(offsets_from_sorted_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, s, keys_out, arcs, n, toffsets);
    return hipGetLastError();
}

}  // namespace bvg
