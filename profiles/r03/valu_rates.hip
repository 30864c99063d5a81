// Issue cost of the VALU instructions the scan kernel leans on, measured on the card: 8 independent chains per lane, 4 waves per SIMD
// (enough to cover the dependent-issue latency), cycles per wave-instruction per SIMD = busy cycles / instructions issued on it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates profiles/r03/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHAINS 8
#define ITERS 32768

#define KERNEL(NAME, ASM, ...)                                                             \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed, uint64_t* cyc) {           \
        uint32_t x[CHAINS];                                                                               \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) x[c] = seed + threadIdx.x * 977u + c * 131u;    \
        uint32_t k = seed | 1u; uint64_t w = seed; (void)w;                                               \
        const uint64_t t0 = clock64();                                                                    \
        for (int i = 0; i < ITERS; i++) {                                                                 \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(x[c]) : "v"(k) __VA_ARGS__);   \
        }                                                                                                 \
        const uint64_t t1 = clock64();                                                                    \
        uint32_t s = 0;                                                                                   \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) s ^= x[c];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                   \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                  \
    }

KERNEL(k_add,      "v_add_u32 %0, %0, %1", )
KERNEL(k_xor,      "v_xor_b32 %0, %0, %1", )
KERNEL(k_mul_lo,   "v_mul_lo_u32 %0, %0, %1", )
KERNEL(k_mul_hi,   "v_mul_hi_u32 %0, %0, %1", )
KERNEL(k_mul_u24,  "v_mul_u32_u24 %0, %0, %1", )
KERNEL(k_mad_u24,  "v_mad_u32_u24 %0, %0, %1, %1", )
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1", )
KERNEL(k_add3,     "v_add3_u32 %0, %0, %1, %1", )
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %1", )
KERNEL(k_bfe,      "v_bfe_u32 %0, %0, 3, 17", )
KERNEL(k_ffbh,     "v_ffbh_u32 %0, %0", )
KERNEL(k_cndmask,  "v_cndmask_b32 %0, %0, %1, vcc", )
KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", : "s20", "s21")
KERNEL(k_cmp_cnd,  "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", : "vcc")
KERNEL(k_cmp_cnd_s, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]", : "s20", "s21")
KERNEL(k_min,      "v_min_u32 %0, %0, %1", )
KERNEL(k_cmpx,     "v_cmp_lt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %1\n s_mov_b64 exec, s[20:21]", : "s20", "s21", "vcc")
KERNEL(k_lshlrev,  "v_lshlrev_b32 %0, %1, %0", )
KERNEL(k_and_or,   "v_and_or_b32 %0, %0, %1, %1", )
KERNEL(k_xad,      "v_xad_u32 %0, %0, %1, %1", )
KERNEL(k_bcnt,     "v_bcnt_u32_b32 %0, %0, %1", )
KERNEL(k_mbcnt,    "v_mbcnt_lo_u32_b32 %0, %0, %1", )
KERNEL(k_cmp,      "v_cmp_lt_u32 vcc, %0, %1", )
KERNEL(k_dpp,      "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", )
KERNEL(k_bperm,    "ds_bpermute_b32 %0, %1, %0", )
KERNEL(k_swizzle,  "ds_swizzle_b32 %0, %0 offset:0x041F", )
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 5\n v_add_u32 %0, s20, %0", : "s20")

// 64-bit operand forms
#define KERNEL64(NAME, ASM)                                                                               \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed, uint64_t* cyc) {           \
        uint64_t x[CHAINS];                                                                               \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) x[c] = ((uint64_t)seed << 20) + threadIdx.x * 977u + c * 131u; \
        uint32_t k = seed | 1u;                                                                           \
        const uint64_t t0 = clock64();                                                                    \
        for (int i = 0; i < ITERS; i++) {                                                                 \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(x[c]) : "v"(k));    \
        }                                                                                                 \
        const uint64_t t1 = clock64();                                                                    \
        uint64_t s = 0;                                                                                   \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) s ^= x[c];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));                           \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                  \
    }
KERNEL64(k_mad_u64,   "v_mad_u64_u32 %0, vcc, %1, %1, %0")
KERNEL64(k_lshl_b64,  "v_lshlrev_b64 %0, %1, %0")
KERNEL64(k_lshr_b64,  "v_lshrrev_b64 %0, %1, %0")
KERNEL64(k_lshl_add64, "v_lshl_add_u64 %0, %0, 1, %0")

typedef void (*kern_t)(uint32_t*, uint32_t, uint64_t*);
struct Case { const char* name; kern_t k; int per; };

int main() {
    const int blocks = 256 * 4, threads = 256;       // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, sizeof(uint32_t) * blocks * threads); hipMalloc(&cyc, sizeof(uint64_t) * blocks);
    std::vector<Case> cases = {
        {"v_add_u32", k_add, 1}, {"v_xor_b32", k_xor, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1}, {"v_mul_u32_u24", k_mul_u24, 1},
        {"v_mad_u32_u24", k_mad_u24, 1}, {"v_lshl_add_u32", k_lshl_add, 1}, {"v_add3_u32", k_add3, 1}, {"v_alignbit_b32", k_alignbit, 1},
        {"v_bfe_u32", k_bfe, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_cndmask_b32 vcc", k_cndmask, 1}, {"v_cndmask_b32 sgpr", k_cndmask_s, 1}, {"v_cmp+v_cndmask vcc", k_cmp_cnd, 2}, {"v_cmp+v_cndmask sgpr", k_cmp_cnd_s, 2}, {"v_min_u32", k_min, 1}, {"cmp+saveexec+add+restore", k_cmpx, 4}, {"v_lshlrev_b32", k_lshlrev, 1}, {"v_and_or_b32", k_and_or, 1},
        {"v_xad_u32", k_xad, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_mbcnt_lo", k_mbcnt, 1}, {"v_cmp_lt_u32", k_cmp, 1}, {"v_add_u32_dpp", k_dpp, 1},
        {"ds_bpermute_b32", k_bperm, 1}, {"ds_swizzle_b32", k_swizzle, 1}, {"v_readlane+v_add", k_readlane, 2},
        {"v_mad_u64_u32", k_mad_u64, 1}, {"v_lshlrev_b64", k_lshl_b64, 1}, {"v_lshrrev_b64", k_lshr_b64, 1}, {"v_lshl_add_u64", k_lshl_add64, 1},
    };
    std::vector<uint64_t> h(blocks);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_add, dim3(blocks), dim3(threads), 0, 0, out, 12345u, cyc);   // clocks up
    hipDeviceSynchronize();
    for (auto& c : cases) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, out, 12345u, cyc);      // warm
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, out, 12345u, cyc);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, sizeof(uint64_t) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
        // a SIMD holds 4 waves here: each issues ITERS * CHAINS * per instructions in `avg` clock64 ticks (100 MHz counter on gfx9: report the event time too)
        const double instr_per_simd = 4.0 * ITERS * CHAINS * c.per;
        const double ns_per_instr = (double)ms * 1e6 / instr_per_simd;                       // whole launch = every SIMD does the same in parallel
        printf("%-26s %8.3f ms   %6.3f ns per wave-instruction per SIMD   (clock64 ticks per block %.0f)\n", c.name, ms, ns_per_instr, avg);
    }
    return 0;
}
