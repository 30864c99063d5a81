// tests/emu/hip/hip_runtime.h — TEST INFRASTRUCTURE, not product code.
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile the product's .hip sources for the HOST, so that the wavefront kernels can
// be stepped through on a CPU (this container has no GPU): asserts, gdb, AddressSanitizer / UBSan on the LDS indexing, printf.  Every
// workgroup runs as `blockDim.x` cooperative fibers on one OS thread (emu.cpp); each cross-lane operation (ballot, shuffle, readlane,
// DPP, wave barrier, __syncthreads) is a rendezvous of the wavefront's (workgroup's) fibers.  Between two rendezvous a lane runs alone,
// in lane order -- or in reverse order with BVG_EMU_ORDER=rev: results that differ between the two orders mean an LDS dependency without a
// wave_sync() between writer and reader.
//
// The product library is NEVER built from this: webgraph-big_amd/Makefile compiles for gfx950 with hipcc only, and the product fails
// loudly without a GPU (tests/test_abi.py).  tests/emu/Makefile builds tests/emu/libbvgraph_emu.so, which only tests load, by name.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>

#define BVG_EMU 1
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }

namespace emu {
struct Idx { unsigned x, y, z; };
extern Idx g_thread, g_block, g_grid, g_bdim;
unsigned char* dyn_lds();                       // the workgroup's dynamic LDS (160 KB, poisoned between launches)
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t dyn_bytes);
// rendezvous of the calling lane's wavefront; values exchanged through a per-wave buffer
uint64_t wave_gather(uint64_t v, const uint64_t** all);   // publishes v, waits for every live lane of the wavefront, returns the lane's own v; *all = the 64 values
void wave_release();                                      // second half of an exchange: the buffer may be reused once every lane has read it
void wave_barrier();
void block_barrier();
uint64_t clock();
}
#define threadIdx (emu::g_thread)
#define blockIdx (emu::g_block)
#define gridDim (emu::g_grid)
#define blockDim (emu::g_bdim)

// ---- runtime API (host side): device memory is host memory, streams are synchronous ----
typedef int hipError_t;
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
hipError_t emu_malloc(void** p, size_t n);
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return emu_malloc((void**)p, n); }
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return emu_malloc((void**)p, n); }
hipError_t hipFree(void* p);
hipError_t hipHostFree(void* p);
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
struct emu_event { uint64_t t; };
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)calloc(1, sizeof(emu_event)); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr);
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)8 << 30; *t = (size_t)16 << 30; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, dyn, stream, ...) emu::launch([=]() { kernel(__VA_ARGS__); }, (grid), (block), (size_t)(dyn))

// ---- device builtins ----
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline long long clock64() { return (long long)emu::clock(); }
inline void __syncthreads() { emu::block_barrier(); }
inline void __threadfence() {}
#define __ATOMIC_RELAXED_EMU 0
#define __HIP_MEMORY_SCOPE_WAVEFRONT 1
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
inline void __builtin_amdgcn_fence(int, const char*) {}
inline void __builtin_amdgcn_wave_barrier() { emu::wave_barrier(); }
inline void __builtin_amdgcn_s_setprio(int) {}
inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u)); }   // v_alignbit_b32
inline void __builtin_amdgcn_s_sleep(int) {}

template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o | v; return o; }
inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
template <typename T> inline T atomicXor(T* p, T v) { T o = *p; *p = o ^ v; return o; }
inline unsigned atomicXor(unsigned* p, unsigned v) { unsigned o = *p; *p = o ^ v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

inline unsigned long long __ballot(int p) {
    const uint64_t* all; emu::wave_gather(p ? 1u : 0u, &all);
    unsigned long long m = 0; for (int i = 0; i < 64; i++) m |= (unsigned long long)(all[i] & 1u) << i;
    emu::wave_release(); return m;
}
template <typename T> inline T emu_lane_read(T v, int src) {
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    uint64_t bits = 0; memcpy(&bits, &v, sizeof(T));
    const uint64_t* all; emu::wave_gather(bits, &all);
    const uint64_t r = all[src & 63];
    emu::wave_release();
    T out; memcpy(&out, &r, sizeof(T)); return out;
}
template <typename T> inline T __shfl(T v, int src, int = 64) { return emu_lane_read(v, src); }
template <typename T> inline T __shfl_xor(T v, int mask, int = 64) { return emu_lane_read(v, (int)((threadIdx.x & 63u) ^ (unsigned)mask)); }
template <typename T> inline T __shfl_up(T v, unsigned delta, int = 64) { const int l = (int)(threadIdx.x & 63u); return emu_lane_read(v, l >= (int)delta ? l - (int)delta : l); }
template <typename T> inline T __shfl_down(T v, unsigned delta, int = 64) { const int l = (int)(threadIdx.x & 63u); return emu_lane_read(v, l + (int)delta < 64 ? l + (int)delta : l); }
inline int __builtin_amdgcn_readlane(int v, int lane) { return emu_lane_read(v, lane); }
// ds_permute_b32 (forward permute): every lane sends `data` to lane (addr / 4) % 64; a lane nobody sends to reads 0, of several senders the highest lane wins
inline int __builtin_amdgcn_ds_permute(int addr, int data) {
    const uint64_t* all; emu::wave_gather(((uint64_t)(uint32_t)addr << 32) | (uint32_t)data, &all);
    const unsigned l = threadIdx.x & 63u; int r = 0;
    for (int j = 0; j < 64; j++) if ((((uint32_t)(all[j] >> 32)) >> 2 & 63u) == l) r = (int)(uint32_t)all[j];
    emu::wave_release();
    return r;
}
// v_mbcnt_lo / _hi: set bits of the mask word below this lane (+ add); inverse ballot: this lane's bit of a wave-uniform mask
inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t m, uint32_t add) { const unsigned l = threadIdx.x & 63u; return add + (uint32_t)__builtin_popcount(l >= 32 ? m : (m & ((1u << l) - 1u))); }
inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t m, uint32_t add) { const unsigned l = threadIdx.x & 63u; return add + (l > 32 ? (uint32_t)__builtin_popcount(m & ((1u << (l - 32)) - 1u)) : 0u); }
inline bool __builtin_amdgcn_inverse_ballot_w64(uint64_t m) { return (m >> (threadIdx.x & 63u)) & 1ull; }
inline int __builtin_amdgcn_readfirstlane(int v) { return emu_lane_read(v, 0); }     // (every kernel here calls it with all lanes active)
// v_mov_b32 with a DPP control as bvg_device.h uses it: row_shr:n (0x110 + n), row_bcast:15 (0x142), row_bcast:31 (0x143); lanes that are masked off
// (row_mask / bank_mask) or have no source keep `old` (bound_ctrl = false)
inline int __builtin_amdgcn_update_dpp(int old, int v, int ctrl, int row_mask, int bank_mask, bool) {
    const uint64_t* all; emu::wave_gather((uint32_t)v, &all);
    const int l = (int)(threadIdx.x & 63u), row = l >> 4, pos = l & 15;
    int r = old;
    if (((row_mask >> row) & 1) && ((bank_mask >> (pos >> 2)) & 1)) {
        if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; if (pos >= n) r = (int)(uint32_t)all[l - n]; }
        else if (ctrl == 0x142) { if (row >= 1) r = (int)(uint32_t)all[row * 16 - 1]; }
        else if (ctrl == 0x143) { if (row >= 2) r = (int)(uint32_t)all[31]; }
        else { fprintf(stderr, "emu: DPP control 0x%x not modelled\n", ctrl); abort(); }
    }
    emu::wave_release();
    return r;
}
using std::max;
using std::min;
inline uint32_t max(uint32_t a, int b) { return a > (uint32_t)b ? a : (uint32_t)b; }
