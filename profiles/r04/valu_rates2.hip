// (round 4: more instructions -- which ones issue at the double rate v_add_u32 / v_xor_b32 showed in round 3 -- and LDS instruction throughput)
// Issue cost of the VALU instructions the scan kernel leans on, measured on the card: 8 independent chains per lane, 4 waves per SIMD
// (enough to cover the dependent-issue latency), cycles per wave-instruction per SIMD = busy cycles / instructions issued on it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates2 profiles/r04/valu_rates2.hip && /tmp/valu_rates2
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

KERNEL(k_add, "v_add_u32 %0, %0, %1", )
KERNEL(k_sub, "v_sub_u32 %0, %0, %1", )
KERNEL(k_and, "v_and_b32 %0, %0, %1", )
KERNEL(k_or, "v_or_b32 %0, %0, %1", )
KERNEL(k_xor, "v_xor_b32 %0, %0, %1", )
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0", )
KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0", )
KERNEL(k_lshlv, "v_lshlrev_b32 %0, %1, %0", )
KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0", )
KERNEL(k_max, "v_max_u32 %0, %0, %1", )
KERNEL(k_min, "v_min_u32 %0, %0, %1", )
KERNEL(k_not, "v_not_b32 %0, %0", )
KERNEL(k_mov, "v_mov_b32 %0, %0", )
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %1", )
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %1", )
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %1", )
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1", )
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 1", )
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1", )
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %1", )
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %1", )
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 17", )
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %1", )
KERNEL(k_ffbh, "v_ffbh_u32 %0, %0", )
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1", )
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", )
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1", )
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %1", )
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %1", )
KERNEL(k_add_f32, "v_add_f32 %0, %0, %1", )
KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1", )
KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0", )
KERNEL(k_min3, "v_min3_u32 %0, %0, %1, %1", )
KERNEL(k_med3, "v_med3_u32 %0, %0, %1, %1", )
KERNEL(k_sad, "v_sad_u32 %0, %0, %1, %1", )
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %1", : "vcc")
KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc", : "vcc")
KERNEL(k_cmp_eq, "v_cmp_eq_u32 vcc, %0, %1", : "vcc")
KERNEL(k_cmp_lt_s, "v_cmp_lt_u32_e64 s[20:21], %0, %1", : "s20", "s21")
KERNEL(k_cnd_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", : "s20", "s21")
KERNEL(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD src0_sel:DWORD src1_sel:WORD_0", )
KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", )
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 5", : "s20")
KERNEL(k_readfirst, "v_readfirstlane_b32 s20, %0", : "s20")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1", )
KERNEL(k_pk_lshl_b16, "v_pk_lshlrev_b16 %0, %1, %0", )
KERNEL(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1", )
KERNEL(k_bperm, "ds_bpermute_b32 %0, %1, %0", )
KERNEL(k_swizzle, "ds_swizzle_b32 %0, %0 offset:0x041F", )
KERNEL(k_mix_mul_add, "v_mul_lo_u32 %0, %0, %1\n v_add_u32 %0, %0, %1", )
KERNEL(k_mix_lshl_xor, "v_lshlrev_b32 %0, 3, %0\n v_xor_b32 %0, %0, %1", )

// LDS throughput: 8 independent accesses in flight per lane, then one wait; addresses: consecutive dwords per lane (conflict-free)
#define LDSKERNEL(NAME, DECL, ASM, NREG)                                                                    \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed, uint64_t* cyc) {           \
        __shared__ uint32_t buf[256 * 4 * CHAINS + 64];                                                   \
        for (int i = threadIdx.x; i < 256 * 4 * CHAINS + 64; i += 256) buf[i] = seed + i;                 \
        __syncthreads();                                                                                  \
        DECL v[CHAINS]; uint32_t addr[CHAINS];                                                            \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) addr[c] = (uint32_t)(size_t)&buf[0] + (c * 256 + threadIdx.x) * 4 * NREG; \
        uint32_t s = 0;                                                                                   \
        const uint64_t t0 = clock64();                                                                    \
        for (int i = 0; i < ITERS / 4; i++) {                                                             \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "=v"(v[c]) : "v"(addr[c]));  \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
            _Pragma("unroll") for (int c = 0; c < CHAINS; c++) s ^= *reinterpret_cast<uint32_t*>(&v[c]);   \
        }                                                                                                 \
        const uint64_t t1 = clock64();                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                   \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                  \
    }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
LDSKERNEL(k_ds_read_b32, uint32_t, "ds_read_b32 %0, %1", 1)
LDSKERNEL(k_ds_read_b64, u32x2, "ds_read_b64 %0, %1", 2)
LDSKERNEL(k_ds_read_b128, u32x4, "ds_read_b128 %0, %1", 4)
LDSKERNEL(k_ds_read2_b32, u32x2, "ds_read2_b32 %0, %1 offset1:1", 2)
typedef void (*kern_t)(uint32_t*, uint32_t, uint64_t*);
struct Case { const char* name; kern_t k; int per; };

int main() {
    const int blocks = 256 * 4, threads = 256;       // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, sizeof(uint32_t) * blocks * threads); hipMalloc(&cyc, sizeof(uint64_t) * blocks);
    std::vector<Case> cases = {{"v_add_u32", k_add, 1}, {"v_sub_u32", k_sub, 1}, {"v_and_b32", k_and, 1}, {"v_or_b32", k_or, 1}, {"v_xor_b32", k_xor, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_lshlrev_b32", k_lshlv, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_max_u32", k_max, 1}, {"v_min_u32", k_min, 1}, {"v_not_b32", k_not, 1}, {"v_mov_b32", k_mov, 1}, {"v_bfi_b32", k_bfi, 1}, {"v_perm_b32", k_perm, 1}, {"v_or3_b32", k_or3, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_add_lshl_u32", k_add_lshl, 1}, {"v_lshl_add_u32", k_lshl_add, 1}, {"v_add3_u32", k_add3, 1}, {"v_and_or_b32", k_and_or, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_bcnt_u32_b32", k_bcnt, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_add_f32", k_add_f32, 1}, {"v_mul_f32", k_mul_f32, 1}, {"v_cvt_f32_u32", k_cvt_f32_u32, 1}, {"v_min3_u32", k_min3, 1}, {"v_med3_u32", k_med3, 1}, {"v_sad_u32", k_sad, 1}, {"v_add_co_u32", k_add_co, 1}, {"v_addc_co_u32", k_addc, 1}, {"v_cmp_eq_u32", k_cmp_eq, 1}, {"v_cmp_lt_u32_e64", k_cmp_lt_s, 1}, {"v_cndmask_b32_e64", k_cnd_s, 1}, {"v_sub_u32_sdwa", k_sub_sdwa, 1}, {"v_mov_b32_dpp", k_dpp_mov, 1}, {"v_readlane_b32", k_readlane, 1}, {"v_readfirstlane_b32", k_readfirst, 1}, {"v_pk_add_u16", k_pk_add_u16, 1}, {"v_pk_lshlrev_b16", k_pk_lshl_b16, 1}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1}, {"ds_bpermute_b32", k_bperm, 1}, {"ds_swizzle_b32", k_swizzle, 1}, {"v_mul_lo+v_add (2 instr)", k_mix_mul_add, 2}, {"v_lshl+v_xor (2 instr)", k_mix_lshl_xor, 2}};
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
    struct LCase { const char* name; kern_t k; };
    std::vector<LCase> lcases = {{"ds_read_b32", k_ds_read_b32}, {"ds_read_b64", k_ds_read_b64}, {"ds_read_b128", k_ds_read_b128}, {"ds_read2_b32", k_ds_read2_b32}};
    for (auto& c : lcases) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, out, 12345u, cyc);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 0, 0, out, 12345u, cyc);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 4 workgroups x 4 waves x (ITERS / 4) x CHAINS LDS instructions (+ as many v_xor, 1.07 ns each per SIMD)
        const double per_cu = 16.0 * (ITERS / 4) * CHAINS;
        printf("%-26s %8.3f ms   %6.3f ns per wave-instruction per CU (includes one v_xor per read on the SIMDs)\n", c.name, ms, (double)ms * 1e6 / per_cu);
    }
    return 0;
}
