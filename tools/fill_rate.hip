// Per-CU fill rate micro-benchmark: LDS-DMA (global_load_lds_dwordx4) vs plain global_load_dwordx4 into VGPRs, with the GEMM's
// access pattern (a wave instruction = 8 rows x 128 B of a row-major bf16 matrix with row stride K), L2-warm data.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_build/fill_rate tools/fill_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ void glds16(const uint16_t *g, uint16_t *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// mode 0: DMA, PER instructions per wave per tile; mode 1: VGPR loads, PER per lane per tile
template <int MODE, int PER>
__global__ __launch_bounds__(256) void fill_kernel(const uint16_t *W, int K, int rows_per_wg, int nk, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint16_t *base = W + (size_t)(blockIdx.x & 3) * rows_per_wg * K;  // 4 slabs shared by all workgroups: L2-resident
    unsigned acc = 0;
    const uint16_t *src[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int row = (wave * PER + i) * 8 + (lane >> 3);
        src[i] = base + (size_t)row * K + (lane & 7) * 8;
    }
    for (int kt = 0; kt < nk; ++kt) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < PER; ++i) glds16(src[i] + (size_t)kt * 64, smem + ((kt & 3) * 4 * PER + wave * PER + i) * 512);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
        } else {
            uint4 v[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) v[i] = *reinterpret_cast<const uint4 *>(src[i] + (size_t)kt * 64);
#pragma unroll
            for (int i = 0; i < PER; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        }
    }
    if (MODE == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc = smem[tid];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int PER>
static void run(const char *name, const uint16_t *W, int K, unsigned *sink)
{
    const int rows = 4 * PER * 8, nk = K / 64, wgs = 256;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const size_t lds = MODE == 0 ? 4 * 4 * PER * 1024 : 1024;
    hipFuncSetAttribute((const void *)fill_kernel<MODE, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill_kernel<MODE, PER>), dim3(wgs), dim3(256), lds, 0, W, K, rows, nk, sink);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fill_kernel<MODE, PER>), dim3(wgs), dim3(256), lds, 0, W, K, rows, nk, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)wgs * rows * K * 2, us = ms * 1e3 / reps;
    printf("%-34s K=%5d  %8.1f us  %7.1f GB/s/CU  (%5.1f B/clk/CU at 2.4 GHz)  total %.2f TB/s\n", name, K, us,
           bytes / wgs / us / 1e3, bytes / wgs / (us * 2400.0), bytes / us / 1e6);
}

int main()
{
    const int K = 8192, rows = 256 * 4 * 6 * 8;
    uint16_t *W; unsigned *sink;
    hipMalloc(&W, (size_t)rows * K * 2); hipMemset(W, 1, (size_t)rows * K * 2); hipMalloc(&sink, 4);
    run<0, 2>("LDS-DMA, 2 pieces/wave/tile", W, K, sink);
    run<0, 4>("LDS-DMA, 4 pieces/wave/tile", W, K, sink);
    run<0, 6>("LDS-DMA, 6 pieces/wave/tile", W, K, sink);
    run<1, 2>("VGPR loads, 2/lane/tile", W, K, sink);
    run<1, 4>("VGPR loads, 4/lane/tile", W, K, sink);
    run<1, 6>("VGPR loads, 6/lane/tile", W, K, sink);
    return 0;
}
