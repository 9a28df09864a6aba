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

// ---------------------------------------------------------------------------------------------------------------
// out[b][n] = bias[n] + sum_k shift_b[k] W[n][k] for the two projections behind a block's folded modulated pre-norms (which = 0:
// qkv, N0 rows, shift row 0 of the block's modulation; which = 1: fc1, N1 rows, shift row `shift_which_off` further): the bias rows
// GaGemmArgs.bias_stride reads (include/ga_dit.h).
// Mapping: shift-stationary, weight-streaming.  A workgroup owns kSbRows groups of 8 weight rows of one projection; wave w owns the
// K-tiles w, w + waves, ... and keeps its 8 shift values per tile and batch item in registers (lane l: row l >> 3, 16-byte chunk l & 7
// of the 8 x 64 tile -- one 1-KiB piece of the tiled weight image per wave load).  All kSbRows tile loads of a wave are in flight
// together; the 8 lanes of a row add up by DPP, the waves' partial sums meet in LDS (fixed order: deterministic).
struct ShiftBiasJob {
    const uint16_t *W[2];
    const float *bias[2];
    const float *shift;      // [B] rows, shift_batch_stride apart
    float *out;              // [B x N0 | B x N1]
    long long shift_batch_stride, shift_which_off;
    int N0, N1, K, B, tiled;
};
constexpr int kSbRows = 16;          // 8-row groups per workgroup
constexpr int kSbTilesPerWave = 4;   // K / 64 <= kSbTilesPerWave * waves of the workgroup (host-checked)
constexpr int kSbLdsFloats = kSbRows * 16 * 8 * 2;   // partial sums of up to 16 waves

__host__ __device__ __forceinline__ int shift_bias_wgs(int N0, int N1)
{
    return (N0 / 8 + kSbRows - 1) / kSbRows + (N1 / 8 + kSbRows - 1) / kSbRows;
}

// PAIRS: batch pairs served by one pass over the weights (LDS: PAIRS * kSbLdsFloats floats); larger batches take further passes
template <int PAIRS>
__device__ __forceinline__ void shift_bias_block(const ShiftBiasJob &j, int wg, float *lds)
{
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wgs0 = (j.N0 / 8 + kSbRows - 1) / kSbRows;
    const int which = wg >= wgs0;
    const int N = which ? j.N1 : j.N0, rg0 = (which ? wg - wgs0 : wg) * kSbRows;
    if (rg0 >= (N >> 3)) return;                                       // workgroup-uniform
    const int nr = min(kSbRows, (N >> 3) - rg0), K = j.K, nk = K >> 6;
    const int row = lane >> 3, ch = lane & 7;
    const float *bias = j.bias[which];
    float *out = j.out + (which ? (size_t)j.B * j.N0 : 0);
    // byte offsets from the (scalar) weight base fit 32 bits: scalar base + per-lane offset addressing, no 64-bit address per load
    const uint32_t lane_off = j.tiled ? (uint32_t)lane * 16u : ((uint32_t)row * K + ch * 8) * 2u;
    const uint32_t rg_step = j.tiled ? (uint32_t)nk * 1024u : (uint32_t)K * 16u, kt_step = j.tiled ? 1024u : 128u;
    const char *wb = reinterpret_cast<const char *>(j.W[which]);
    const size_t pair_floats = (size_t)kSbRows * nw * 16;              // lds[pair][row group][wave][row][2]
    for (int b0 = 0; b0 < j.B; b0 += 2 * PAIRS) {
        const int npairs = min(PAIRS, (j.B - b0 + 1) >> 1);
        const float *sh = j.shift + (size_t)b0 * j.shift_batch_stride + (which ? j.shift_which_off : 0) + ch * 8;
        __syncthreads();                                               // (the previous pass's sums have been read)
#pragma unroll 1
        for (int i = 0; i < kSbTilesPerWave; ++i) {
            const int kt = w + i * nw;
            if (kt >= nk) break;                                       // wave-uniform
#pragma unroll 1
            for (int h = 0; h < kSbRows; h += 8) {                     // 8 tile loads in flight per lane (the register budget of the hosts)
                uint4 wv[8];
                const uint32_t off0 = (uint32_t)rg0 * rg_step + (uint32_t)kt * kt_step + lane_off;
#pragma unroll
                for (int r = 0; r < 8; ++r) wv[r] = *reinterpret_cast<const uint4 *>(wb + (off0 + (uint32_t)min(h + r, nr - 1) * rg_step));
#pragma unroll 1
                for (int bp = 0; bp < npairs; ++bp) {                  // every batch pair of the pass from the same registers
                    const float *s0 = sh + (size_t)(2 * bp) * j.shift_batch_stride + kt * 64;
                    const float *s1 = (b0 + 2 * bp + 1 < j.B) ? s0 + j.shift_batch_stride : s0;
                    const float4 a0 = *reinterpret_cast<const float4 *>(s0), a1 = *reinterpret_cast<const float4 *>(s0 + 4);
                    const float4 c0 = *reinterpret_cast<const float4 *>(s1), c1 = *reinterpret_cast<const float4 *>(s1 + 4);
                    float *pl = lds + bp * pair_floats;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        uint4 x = wv[r];
                        asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w));   // (unpack per pair: hoisted out of the bp loop the 64 floats spill)
                        const float w0 = __uint_as_float(x.x << 16), w1 = __uint_as_float(x.x & 0xFFFF0000u), w2 = __uint_as_float(x.y << 16),
                                    w3 = __uint_as_float(x.y & 0xFFFF0000u), w4 = __uint_as_float(x.z << 16), w5 = __uint_as_float(x.z & 0xFFFF0000u),
                                    w6 = __uint_as_float(x.w << 16), w7 = __uint_as_float(x.w & 0xFFFF0000u);
                        float v0 = (w0 * a0.x + w1 * a0.y) + (w2 * a0.z + w3 * a0.w) + (w4 * a1.x + w5 * a1.y) + (w6 * a1.z + w7 * a1.w);
                        float v1 = (w0 * c0.x + w1 * c0.y) + (w2 * c0.z + w3 * c0.w) + (w4 * c1.x + w5 * c1.y) + (w6 * c1.z + w7 * c1.w);
                        v0 += __shfl_xor(v0, 1, 64); v0 += __shfl_xor(v0, 2, 64); v0 += __shfl_xor(v0, 4, 64);
                        v1 += __shfl_xor(v1, 1, 64); v1 += __shfl_xor(v1, 2, 64); v1 += __shfl_xor(v1, 4, 64);
                        if (ch == 0) {                                 // this wave's own slot: its K-tiles add up here
                            float2 *slot = reinterpret_cast<float2 *>(pl + (((size_t)(h + r) * nw + w) * 8 + row) * 2);
                            const float2 prev = i ? *slot : make_float2(0.f, 0.f);
                            *slot = make_float2(prev.x + v0, prev.y + v1);
                        }
                    }
                }
            }
        }
        if (w >= nk && lane < 16) {                                    // a wave without a K-tile: zero partial sums
            for (int bp = 0; bp < npairs; ++bp)
                for (int r = 0; r < kSbRows; ++r) lds[bp * pair_floats + ((size_t)r * nw + w) * 16 + lane] = 0.f;
        }
        __syncthreads();
        for (int o = threadIdx.x; o < npairs * nr * 16; o += blockDim.x) {
            const int bp = o / (nr * 16), q = o - bp * (nr * 16), r = q >> 4, rw = (q >> 1) & 7, bb = q & 1;
            const int bi = b0 + 2 * bp + bb;
            if (bi >= j.B) continue;
            float sum = 0.f;
            for (int ww = 0; ww < nw; ++ww) sum += lds[bp * pair_floats + (((size_t)r * nw + ww) * 8 + rw) * 2 + bb];
            const int n = (rg0 + r) * 8 + rw;
            out[(size_t)bi * N + n] = (bias ? bias[n] : 0.f) + sum;
        }
    }
}

// Round 6: weight ranges the idle workgroups of an attention launch pull towards the Infinity Cache for a GEMM that runs a few launches
// later (fc2's 8 MB: 24.0 -> 21.1 us at 1536 rows, 19.3 -> 15.7 us at 768 when its weights are there instead of in HBM,
// tools/warm_vs_cold.py).  Plain loads whose values are folded into a word nobody reads: nothing depends on them.
constexpr int kPfRanges = 6;
struct PrefetchJob {
    const char *ptr[kPfRanges];      // nullptr: unused slot
    unsigned bytes[kPfRanges];       // multiples of 1024
};
__device__ __forceinline__ void prefetch_block(const PrefetchJob &j, int wg, int nwgs)
{
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
#pragma unroll 1
    for (int r = 0; r < kPfRanges; ++r) {
        if (!j.ptr[r]) continue;
        const unsigned pieces = j.bytes[r] >> 10, per = (pieces + nwgs - 1) / nwgs;     // 1-KiB pieces: one wave load each
        const unsigned p0 = (unsigned)wg * per, p1 = min(p0 + per, pieces);
        const char *base = j.ptr[r] + (size_t)lane * 16;
#pragma unroll 1
        for (unsigned p = p0 + w; p < p1; p += 8 * nw) {                              // eight loads in flight per lane
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const uint4 *>(base + (size_t)min(p + (unsigned)u * nw, p1 - 1) * 1024);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    asm volatile("" ::"v"(acc));   // (keeps the loads)
}

// dit_attention.hip: the attention launch with `tail` workgroups behind its grid that compute one ShiftBiasJob (the self-attention of a
// CFG pair fills 192 of the 256 CUs with one 96-KiB-LDS workgroup each: the job's weight stream runs on the idle ones)
int attention_with_tail(const GaAttentionArgs *a, const ShiftBiasJob *job, void *stream, const PrefetchJob *pf = nullptr, int pf_wgs = 0);
int attention_workgroups(const GaAttentionArgs *a);
bool attention_fuses_q(const GaAttentionArgs *a);
// dit_attention_hd.hip: the V^T variant for head dims other than 64 with prefetch tail workgroups behind its grid
int attention_hd_with_tail(const GaAttentionHdArgs *a, void *stream, const PrefetchJob *pf, int pf_wgs, const ShiftBiasJob *job = nullptr);
int attention_hd_workgroups(const GaAttentionHdArgs *a);
bool attention_hd_hosts_shift_bias(const GaAttentionHdArgs *a, int K);

}  // namespace gadit
