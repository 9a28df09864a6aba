#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, float *o1, float *o2, float *o3) {
    int l = threadIdx.x;
    float a = in[l], b = in[l + 64];
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    o1[l] = __builtin_bit_cast(float, r[0]); o1[l + 64] = __builtin_bit_cast(float, r[1]);
    auto r2 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    o2[l] = __builtin_bit_cast(float, r2[0]); o2[l + 64] = __builtin_bit_cast(float, r2[1]);
    float v = a;
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
    o3[l] = v;
}
int main() {
    float h[128], *d, *o1, *o2, *o3, r1[128], r2[128], r3[64];
    for (int i = 0; i < 64; ++i) { h[i] = i; h[64 + i] = 100 + i; }
    hipMalloc(&d, 512); hipMalloc(&o1, 512); hipMalloc(&o2, 512); hipMalloc(&o3, 256);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o1, o2, o3);
    hipMemcpy(r1, o1, 512, hipMemcpyDeviceToHost); hipMemcpy(r2, o2, 512, hipMemcpyDeviceToHost); hipMemcpy(r3, o3, 256, hipMemcpyDeviceToHost);
    printf("swap32 a:"); for (int i = 0; i < 64; i += 8) printf(" %g", r1[i]); printf("\nswap32 b:"); for (int i = 0; i < 64; i += 8) printf(" %g", r1[64 + i]);
    printf("\nswap16 a:"); for (int i = 0; i < 64; i += 8) printf(" %g", r2[i]); printf("\nswap16 b:"); for (int i = 0; i < 64; i += 8) printf(" %g", r2[64 + i]);
    printf("\nscan4:"); for (int i = 0; i < 20; ++i) printf(" %g", r3[i]); printf("\n");
    return 0;
}
