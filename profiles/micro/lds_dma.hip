// Micro-check of the gfx950 direct-to-LDS load used for stream-window prefetch: global_load_lds_dwordx4 puts lane i's 16 bytes
// at (uniform LDS base) + 16*i.  Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma lds_dma.hip ; run: ./lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* g, unsigned* out, unsigned skew) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lane = threadIdx.x;
    for (unsigned i = lane; i < 1024; i += 64) ((unsigned*)lds)[i] = 0xDEADBEEFu;
    __syncthreads();
    const unsigned char* src = g + skew;                       // 16-byte aligned source
    for (int c = 0; c < 2; c++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds + 256 + c * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);                              // vmcnt(0): data has landed in LDS
    __syncthreads();
    for (unsigned i = lane; i < 1024; i += 64) out[i] = ((unsigned*)lds)[i];
}
int main() {
    std::vector<unsigned char> h(8192); for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 7 + 3);
    unsigned char* d; unsigned* o; hipMalloc(&d, h.size()); hipMalloc(&o, 4096); hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad = 0;
    for (unsigned skew : {0u, 16u, 4080u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, d, o, skew);
        std::vector<unsigned> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 1024; i++) {
            unsigned want = 0xDEADBEEFu;
            if (i >= 64 && i < 64 + 512) { unsigned b = (i - 64) * 4 + skew; want = h[b] | (h[b + 1] << 8) | (h[b + 2] << 16) | ((unsigned)h[b + 3] << 24); }
            if (r[i] != want) { if (bad < 5) printf("skew %u word %d got %08x want %08x\n", skew, i, r[i], want); bad++; }
        }
    }
    printf(bad ? "LDS DMA MISMATCH (%d)\n" : "LDS DMA OK\n", bad);
    return bad != 0;
}
