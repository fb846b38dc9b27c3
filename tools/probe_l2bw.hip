// Probe: per-CU L2 -> LDS (LDS-DMA) and L2 -> VGPR bandwidth with D outstanding 1-KiB loads per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int DEPTH, bool LDSDMA>
__global__ __launch_bounds__(512) void k_bw(const char* __restrict__ src, size_t span, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // every block walks the same window (L2 resident after the first pass): offset by block to spread channels
    size_t off = ((size_t)blockIdx.x * 65536) % span;
    float4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* p = src + (off + (size_t)(wave * DEPTH + d) * 1024 + lane * 16) % span;
            if constexpr (LDSDMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(smem + (wave * DEPTH + d) * 1024), 16, 0, 0);
            } else {
                const float4 v = *reinterpret_cast<const float4*>(p);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        if constexpr (LDSDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        off = (off + 8 * DEPTH * 1024) % span;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.f) sink[0] = acc.x;
    if (LDSDMA && smem[threadIdx.x] == 77) sink[1] = 1.f;
}

template <int DEPTH, bool LDSDMA>
int run(const char* src, size_t span, float* sink, const char* name) {
    const int iters = 2000, blocks = 256;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto kern = k_bw<DEPTH, LDSDMA>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 8 * DEPTH * 1024, 0, src, span, 50, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 8 * DEPTH * 1024, 0, src, span, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double bytes = (double)blocks * iters * 8 * DEPTH * 1024;
    printf("%-28s span %6.1f MB depth %d: %.2f TB/s chip, %.1f GB/s per CU (%.1f B/clk @2.4GHz)\n", name, span / 1e6, DEPTH,
           bytes / ms / 1e9, bytes / ms / 1e6 / blocks, bytes / ms / 1e6 / blocks / 2.4);
    return 0;
}

int main() {
    char* src; float* sink;
    const size_t big = 512ull << 20;
    CK(hipMalloc(&src, big)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(src, 1, big));
    for (size_t span : {(size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)512 << 20}) {
        run<2, true>(src, span, sink, "LDS-DMA (global_load_lds)");
        run<4, true>(src, span, sink, "LDS-DMA (global_load_lds)");
        run<8, true>(src, span, sink, "LDS-DMA (global_load_lds)");
        run<4, false>(src, span, sink, "global_load_dwordx4 -> VGPR");
        run<8, false>(src, span, sink, "global_load_dwordx4 -> VGPR");
    }
    return 0;
}
