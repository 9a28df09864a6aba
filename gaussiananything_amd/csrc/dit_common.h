// dit_common.h -- shared device helpers of the DiT kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ga_dit.h"

namespace gadit {

using bf16x8 = __attribute__((ext_vector_type(8))) short;   // one MFMA 16x16x32 bf16 operand: 8 bf16 = 4 VGPRs
using f32x4 = __attribute__((ext_vector_type(4))) float;    // one MFMA 16x16 accumulator fragment

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

__device__ __forceinline__ uint16_t f32_to_bf16(float f)  // round to nearest even (NaN kept quiet)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace gadit
