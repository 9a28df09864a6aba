// LDS atomic throughput on gfx950: cycles per wave instruction for ds_add_f32 / ds_add_u32 / ds_write_b32 / ds_pk_add_f16 with
// 64 distinct addresses (stride 19 words, as the backward's gradient image), 8 lanes per address, and one address.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_rate.hip -o tools/_build/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int pattern, int reps, int active)
{
    __shared__ float buf[128 * 19 + 64];
    for (int i = threadIdx.x; i < 128 * 19 + 64; i += 256) buf[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int e = pattern == 0 ? lane : pattern == 1 ? (lane >> 3) : 0;
    e += wave * 7;
    float *p = &buf[e * 19];
    const bool on = lane < active;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            if (on) {
                if (MODE == 0) __hip_atomic_fetch_add(p + q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(p + q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 2) reinterpret_cast<volatile float *>(p)[q] = (float)r;
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (buf[threadIdx.x] == 12345.f) out[0] = 0;
}

int main()
{
    unsigned long long *d;
    hipMalloc(&d, 4096 * 8);
    const char *names[3] = {"ds_add_f32", "ds_add_u32", "ds_write_b32"};
    const char *pats[3] = {"64 addresses", "8 lanes/address", "1 address"};
    for (int mode = 0; mode < 3; ++mode)
        for (int pat = 0; pat < 3; ++pat)
            for (int active : {64, 22, 8}) {
                const int reps = 200, wgs = 256 * 6;
                for (int it = 0; it < 2; ++it) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, d, pat, reps, active);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, d, pat, reps, active);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, d, pat, reps, active);
                    hipDeviceSynchronize();
                }
                std::vector<unsigned long long> h(wgs);
                hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
                double s = 0; for (auto v : h) s += (double)v;
                // 6 workgroups x 4 waves share a CU's LDS: cycles per wave instruction as the CU sees it
                printf("%-13s %-16s active %2d: %.1f cycles/instr/wave (x24 waves per CU -> %.2f CU cycles per instruction)\n", names[mode], pats[pat], active,
                       s / wgs / (reps * 18.0), s / wgs / (reps * 18.0) / 24.0);
            }
    return 0;
}
