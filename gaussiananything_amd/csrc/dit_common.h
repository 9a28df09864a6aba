// dit_common.h -- shared device helpers of the DiT kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ga_dit.h"

namespace gadit {

using bf16x8 = __attribute__((ext_vector_type(8))) short;   // one MFMA 16x16x32 bf16 operand: 8 bf16 = 4 VGPRs
using f32x4 = __attribute__((ext_vector_type(4))) float;    // one MFMA 16x16 accumulator fragment

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

// fp32 -> bf16, round to nearest even: native __bf16 conversions, which hipcc lowers to v_cvt_pk_bf16_f32 on gfx950
// (a hand-rolled bit trick with a NaN test costs ~6 VALU ops AND a divergent branch per element).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace gadit
