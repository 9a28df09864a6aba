// dit_ops.hip -- the HBM-bound row kernels around the DiT GEMMs, gfx950.  Each fuses what the reference runs as several
// elementwise torch ops:
//   rmsnorm_modulate_kernel : RMSNorm (dit/norm.py:29-43) + t2i_modulate (dit_models_xformers.py:53-54) + bf16 cast
//   small_linear_kernel     : the conditioning path's few-row Linear layers with SiLU (TimestepEmbedder :88-128,
//                             pooled_vec_embedder dit_i23d.py:501-509, adaLN_modulation :209-214)
//   timestep_freq_kernel    : sinusoidal features of t (TimestepEmbedder.timestep_embedding)
//   layernorm_rows_kernel   : LayerNorm with affine of the pooled image vector (pooled_vec_embedder.0)
//   mod_table_kernel        : (scale_shift_table[None] + t0.reshape(B,6,-1)) for all blocks at once (:769-770)
//   embed_tokens_kernel     : x_embedder.fc1 + tanh-GELU (timm Mlp, K = 3 or 10) and the NeRF positional encoding +
//                             xyz_projection of stage 2 (vit_triplane.py:187-229, utils/nerf_utils.py:16-66)
//   final_layer_kernel      : T2IFinalLayer (dit_models_xformers.py:62-85): LayerNorm(no affine) + modulate + Linear
// All are one-wave-per-row with 16-byte accesses where the layout allows.
#include <stdlib.h>

#include <algorithm>

#include "dit_common.h"

namespace gadit {

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_modulate_kernel(GaRmsNormArgs a)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.M) return;
    const int D = a.D;
    const float *x = a.x + (size_t)row * D;
    // D <= 2048, D % 4 == 0: lane owns the float4 at d = c*256 + lane*4 of every 256-wide chunk c.  Everything the row
    // needs (x, norm weight, scale, shift) is requested up front: the kernel is one memory round trip plus a wave sum.
    const int b = row / a.rows_per_batch;
    const float *sc = a.scale ? a.scale + (size_t)b * a.mod_stride : nullptr;
    const float *sh = a.shift ? a.shift + (size_t)b * a.mod_stride : nullptr;
    float4 v[8], w[8], s4[8], h4[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int d = c * 256 + lane * 4;
        if (d < D) {
            v[c] = *reinterpret_cast<const float4 *>(x + d);
            if (a.row_bias && row >= a.row_bias_first) {  // wave-uniform
                const float4 rb = *reinterpret_cast<const float4 *>(a.row_bias + d);
                v[c].x += rb.x; v[c].y += rb.y; v[c].z += rb.z; v[c].w += rb.w;
                *reinterpret_cast<float4 *>(const_cast<float *>(x) + d) = v[c];
            }
            w[c] = *reinterpret_cast<const float4 *>(a.weight + d);
            if (sc) {
                s4[c] = *reinterpret_cast<const float4 *>(sc + d);
                h4[c] = *reinterpret_cast<const float4 *>(sh + d);
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c * 256 + lane * 4 < D) ss += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)D + 1e-5f);
    uint16_t *o = a.out + (size_t)row * D;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int d = c * 256 + lane * 4;
        if (d < D) {
            float y[4] = {v[c].x * rs * w[c].x, v[c].y * rs * w[c].y, v[c].z * rs * w[c].z, v[c].w * rs * w[c].w};
            if (sc) {
                y[0] = y[0] * (1.f + s4[c].x) + h4[c].x; y[1] = y[1] * (1.f + s4[c].y) + h4[c].y;
                y[2] = y[2] * (1.f + s4[c].z) + h4[c].z; y[3] = y[3] * (1.f + s4[c].w) + h4[c].w;
            }
            *reinterpret_cast<uint2 *>(o + d) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }

// One wave per output feature n; a lane owns 8 consecutive k of every 512-wide chunk (one 16-byte weight load, two float4
// loads per batch row), so K = 1024 is two independent round trips instead of 16 dependent scalar ones.  K % 8 == 0.
// optional second output of small_linear_kernel: mod[blk][b][n] = tables[blk][n] + y[b][n] for every block (round 5: the adaLN linear of
// the evaluation's head writes the blocks' modulation tables itself; a launch of its own before)
struct ModOut {
    const float *tables[64];
    float *mod;
    int depth;
};

__global__ __launch_bounds__(256) void small_linear_kernel(GaSmallLinearArgs a, ModOut mo)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= a.N) return;
    float acc[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) acc[b] = 0.f;
    const uint16_t *w = a.W + (size_t)n * a.K;
    if (a.act_in == 2) {   // x = timesteps [B]; the input row is TimestepEmbedder.timestep_embedding(t) (K = 256), formed here
        // exactly as timestep_freq_kernel forms it: [cos(t f_j) | sin(t f_j)], f_j = exp(-ln(10000) j / 128)
        const int k = lane * 8;
        if (k < a.K) {
            const uint4 wr = *reinterpret_cast<const uint4 *>(w + k);
            const uint32_t ww[4] = {wr.x, wr.y, wr.z, wr.w};
            float fr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = k + e, f = j < 128 ? j : j - 128;
                fr[e] = expf(-9.210340371976184f * (float)f / 128.0f);
            }
            float accb = 0.f;
#pragma unroll 1
            for (int b = 0; b < a.B; ++b) {   // (not unrolled: 16 x 8 inlined libm sin / cos were 80 KB of code)
                    const float tb = a.x[b];
                    accb = 0.f;
#pragma unroll 1
                    for (int e = 0; e < 8; ++e) {
                        const float arg = tb * fr[e];
                        const float xv = (k + e) < 128 ? cosf(arg) : sinf(arg);
                        const uint32_t we = e < 2 ? ww[0] : (e < 4 ? ww[1] : (e < 6 ? ww[2] : ww[3]));
                        const float wv = (e & 1) ? __uint_as_float(we & 0xffff0000u) : __uint_as_float(we << 16);
                        accb += bf16_to_f32(f32_to_bf16(xv)) * wv;
                    }
                    // acc[] is a register array: select the slot without dynamic indexing
#pragma unroll
                    for (int bb = 0; bb < 16; ++bb) acc[bb] = bb == b ? accb : acc[bb];
            }
        }
    } else
#pragma unroll 2
    for (int k = lane * 8; k < a.K; k += 512) {
        const uint4 wr = *reinterpret_cast<const uint4 *>(w + k);
        const uint32_t ww[4] = {wr.x, wr.y, wr.z, wr.w};
        float wv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wv[2 * e] = __uint_as_float(ww[e] << 16);
            wv[2 * e + 1] = __uint_as_float(ww[e] & 0xffff0000u);
        }
#pragma unroll
        for (int b = 0; b < 16; ++b)
            if (b < a.B) {
                const float4 x0 = *reinterpret_cast<const float4 *>(a.x + (size_t)b * a.K + k);
                const float4 x1 = *reinterpret_cast<const float4 *>(a.x + (size_t)b * a.K + k + 4);
                float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (a.act_in == 1) xv[e] = silu(xv[e]);
                    // the reference runs these Linear layers under bf16 autocast: inputs are rounded to bf16
                    acc[b] += bf16_to_f32(f32_to_bf16(xv[e])) * wv[e];
                }
            }
    }
#pragma unroll
    for (int b = 0; b < 16; ++b)
        if (b < a.B) {
            float v = wave_sum(acc[b]);      // (the butterfly leaves the total in every lane)
            if (a.bias) v += a.bias[n];
            if (a.act_out == 1) v = silu(v);
            if (a.add) v += a.add[(size_t)b * a.N + n];
            if (lane == 0) a.y[(size_t)b * a.N + n] = v;
            // the blocks' modulation tables: lane l serves block l (one load, one store each -- a loop in one lane was 24 dependent round trips)
            for (int blk = lane; blk < mo.depth; blk += 64) mo.mod[((size_t)blk * a.B + b) * a.N + n] = mo.tables[blk][n] + v;
        }
}

__global__ void timestep_freq_kernel(const float *__restrict__ t, float *__restrict__ out, int B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // B * 256
    if (i >= B * 256) return;
    const int b = i >> 8, j = i & 255, half = 128;
    const int f = j < half ? j : j - half;
    const float freq = expf(-9.210340371976184f * (float)f / (float)half);  // ln(10000)
    const float arg = t[b] * freq;
    out[i] = j < half ? cosf(arg) : sinf(arg);
}

__global__ __launch_bounds__(64) void layernorm_rows_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ y,
                                                            int D, float eps)
{
    const int row = blockIdx.x, lane = threadIdx.x;
    const float *xr = x + (size_t)row * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += xr[d];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int d = lane; d < D; d += 64) { const float c = xr[d] - mean; q += c * c; }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    for (int d = lane; d < D; d += 64) y[(size_t)row * D + d] = (xr[d] - mean) * rs * w[d] + bias[d];
}

// mod[blk][b][j][d] = table_blk[j][d] + t0[b][j*D + d]      (j = shift_msa scale_msa gate_msa shift_mlp scale_mlp gate_mlp)
struct ModTableArgs {
    const float *tables[64];
    const float *t0;
    float *mod;
    int depth, B, D;
};

__global__ void mod_table_kernel(ModTableArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_blk = (int64_t)a.B * 6 * a.D;
    if (i >= per_blk * a.depth) return;
    const int blk = (int)(i / per_blk);
    const int64_t r = i - (int64_t)blk * per_blk;
    const int b = (int)(r / (6 * a.D)), jd = (int)(r - (int64_t)b * 6 * a.D);
    a.mod[i] = a.tables[blk][jd] + a.t0[(size_t)b * 6 * a.D + jd];
}

// ---------------------------------------------------------------------------------------------------------------
// The shift of a folded modulated pre-norm, carried through the projection (include/ga_dit.h, GaGemmArgs.bias_stride):
//   out[job][b][n] = bias[n] + sum_k shift_b[k] W[n][k]      for the qkv (which = 0) and fc1 (which = 1) weights of every block
// in ONE launch per evaluation (the modulation of all blocks is known once t is embedded).  A wave owns 8 weight rows -- one 1-KiB
// tile column of the tiled image, K/64 tiles contiguous -- lane l reads the 16 bytes (row l >> 3, chunk l & 7) of every tile,
// so the weights stream through once per batch pair at full request width; 350 MB for DiT-L.
struct ShiftBiasArgs {
    const uint16_t *W[128];     // [block][which]
    const float *bias[128];
    const float *shift;         // mod table [block][B][6][D]; which = 0: row 0 (shift_msa), which = 1: row 3 (shift_mlp)
    float *out;                 // [block][ B x N0 | B x N1 ]
    long long shift_block_stride, shift_batch_stride, shift_which_off, out_block_stride;
    int N0, N1, K, B, jobs, tiled;
};

__global__ __launch_bounds__(1024) void shift_bias_kernel(ShiftBiasArgs a)
{
    constexpr int kPairs = 4;                                          // 8 batch items per pass over the weights: 64 KiB of partial sums
    __shared__ __attribute__((aligned(16))) float sh[kPairs * kSbLdsFloats];
    ShiftBiasJob j;
    j.N0 = a.N0; j.N1 = a.N1; j.K = a.K; j.B = a.B; j.tiled = a.tiled;
    const int wgs = shift_bias_wgs(a.N0, a.N1);                        // workgroups per block of the model (grid = jobs * wgs)
    const int blk = blockIdx.x / wgs;
    j.W[0] = a.W[2 * blk]; j.W[1] = a.W[2 * blk + 1]; j.bias[0] = a.bias[2 * blk]; j.bias[1] = a.bias[2 * blk + 1];
    j.shift = a.shift + (size_t)blk * a.shift_block_stride; j.out = a.out + (size_t)blk * a.out_block_stride;
    j.shift_batch_stride = a.shift_batch_stride; j.shift_which_off = a.shift_which_off;
    shift_bias_block<kPairs>(j, blockIdx.x - blk * wgs, sh);
}

// ---------------------------------------------------------------------------------------------------------------
// h[m][n] = gelu_tanh(sum_c x[m][c] W1[n][c] + b1[n]) as bf16 (A operand of the x_embedder.fc2 GEMM) and, for stage 2,
// xres[m][n] = sum_j PE(xyz[m])[j] Wx[n][j] + bx[n]  (fp32 residual stream start; fc2 is accumulated on top).
struct EmbedArgs {
    int M, D, C, stage2;
    const float *x, *w1, *b1, *xyz, *wx, *bx;
    uint16_t *h;
    float *xres;
};

// Round 6: a workgroup = kET tokens x one 128-column tile of the outputs (grid: token groups x column tiles).  One token per workgroup
// re-read the whole [D, 63] positional-embedding weight (258 KB) per token with a stride of 63 floats between lanes -- 30 us for the 768
// tokens of a stage-2 evaluation.  Now a tile's 128 weight rows are staged once through LDS (coalesced reads of the contiguous rows,
// stored transposed: conflict-free reads with lanes = output columns) and serve kET tokens; the sums run over the features in the same
// order as before (same bits).  (Four tokens x all D columns per workgroup -- 192 workgroups, eight tiles each -- measured 61 us: too
// few waves for a chain of staging rounds.)
constexpr int kET = 8;
__global__ __launch_bounds__(256) void embed_tokens_kernel(EmbedArgs a)
{
    __shared__ float sx[kET][16];
    __shared__ float spe[kET][64];
    __shared__ float wt[63][129];          // [feature][row of the 128-row weight tile] (+1: the transposing stores spread over the banks)
    const int tid = threadIdx.x, m0 = blockIdx.x * kET, n0 = blockIdx.y * 128;
    if (tid < kET * 16) {
        const int t = tid >> 4, c = tid & 15, m = m0 + t;
        sx[t][c] = (c < a.C && m < a.M) ? bf16_to_f32(f32_to_bf16(a.x[(size_t)m * a.C + c])) : 0.f;  // autocast
    }
    if (a.stage2) {
        // [x, sin(2^k x), cos(2^k x)]_{k=0..9}: index 3 + 6k + {0..2 sin, 3..5 cos}
        for (int i = tid; i < kET * 64; i += 256) {
            const int t = i >> 6, j = i & 63, m = m0 + t;
            float v = 0.f;
            if (j < 63 && m < a.M) {
                if (j < 3) v = a.xyz[(size_t)m * 3 + j];
                else {
                    const int k = (j - 3) / 6, r = (j - 3) % 6, c = r % 3;
                    const float arg = a.xyz[(size_t)m * 3 + c] * (float)(1 << k);
                    v = r < 3 ? sinf(arg) : cosf(arg);
                }
            }
            spe[t][j] = bf16_to_f32(f32_to_bf16(v));
        }
        const int rows = min(128, a.D - n0), cnt = rows * 63;
        const float *src = a.wx + (size_t)n0 * 63;
        for (int i = tid; i < cnt; i += 256) {
            const int r = i / 63, j = i - r * 63;
            wt[j][r] = bf16_to_f32(f32_to_bf16(src[i]));
        }
    }
    __syncthreads();
    const int nl = tid & 127, th = tid >> 7, n = n0 + nl;          // my column; my tokens: th, th + 2, th + 4, th + 6
    if (n >= a.D) return;
    float w1r[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) w1r[c] = c < a.C ? bf16_to_f32(f32_to_bf16(a.w1[(size_t)n * a.C + c])) : 0.f;
    const float b1n = a.b1[n], bxn = a.stage2 ? a.bx[n] : 0.f;
#pragma unroll
    for (int q = 0; q < kET / 2; ++q) {
        const int t = th + 2 * q, m = m0 + t;
        if (m >= a.M) continue;
        float acc = b1n;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < a.C) acc += sx[t][c] * w1r[c];
        const float u = 0.7978845608028654f * (acc + 0.044715f * acc * acc * acc);
        a.h[(size_t)m * a.D + n] = f32_to_bf16(0.5f * acc * (1.0f + tanhf(u)));
        float r = 0.f;
        if (a.stage2) {
            r = bxn;
            for (int j = 0; j < 63; ++j) r += spe[t][j] * wt[j][nl];
        }
        a.xres[(size_t)m * a.D + n] = r;
    }
}

// out[m][c] = sum_d (LN(x[m])[d] * (1 + scale[b][d]) + shift[b][d]) * W[c][d] + bias[c];  (shift, scale) = table + t
struct FinalArgs {
    int M, D, Cout, rows_per_batch;
    const float *x, *table, *t, *w, *bias;
    float *out;
    // fused sampler step (GaDitSamplerStep), state == nullptr: plain evaluation
    float cfg_scale;
    int cfg;
    const float *dt;
    float *state, *traj;
    long long traj_stride;
    const int *counter;
};

// One wave per row, the row (D <= 2048, D % 4 == 0) held in registers as in rmsnorm_modulate_kernel: one memory round trip
// for x / table / t, the LayerNorm statistics by two wave sums, then Cout <= 16 dot products against the L2-resident weight.
// The sampler arithmetic of the fused step, every operation rounded once (no contraction into FMAs): these are the very
// roundings of the eager PyTorch loop (forward_with_cfg: u + s * (c - u); euler: y + dt * v).
__device__ __forceinline__ float cfg_combine(float c, float u, float s)
{
    const float d = c - u;
    float m = s * d;
    asm volatile("" : "+v"(m));   // this TU is built with -ffp-contract=fast: keep the product a value of its own
    return u + m;
}
__device__ __forceinline__ float euler_update(float y, float dt, float v)
{
    float p = dt * v;
    asm volatile("" : "+v"(p));
    return y + p;
}

__device__ __forceinline__ void final_layer_row(const FinalArgs &a, int row, int lane, float (&res)[16], const float *wl = nullptr)
{
    const int D = a.D, b = row / a.rows_per_batch;
    const float *x = a.x + (size_t)row * D, *tb = a.t + (size_t)b * D;
    float4 v[8], sh[8], sc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int d = c * 256 + lane * 4;
        if (d < D) {
            v[c] = *reinterpret_cast<const float4 *>(x + d);
            const float4 t4 = *reinterpret_cast<const float4 *>(tb + d);
            const float4 h4 = *reinterpret_cast<const float4 *>(a.table + d);
            const float4 s4 = *reinterpret_cast<const float4 *>(a.table + D + d);
            sh[c] = make_float4(h4.x + t4.x, h4.y + t4.y, h4.z + t4.z, h4.w + t4.w);
            sc[c] = make_float4(s4.x + t4.x, s4.y + t4.y, s4.z + t4.z, s4.w + t4.w);
        }
    }
    // LayerNorm statistics in ONE reduction round (round 5), about a pivot (round 6): sum and sum of squares of (x - x[0]) are reduced
    // together -- the plain E[x^2] - mean^2 cancels when a row carries a common offset much larger than its spread; shifted by an
    // element of the row the two terms are of the size of the spread itself (T2IFinalLayer's LayerNorm is two-pass upstream)
    const float pivot = __shfl(v[0].x, 0, 64);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c * 256 + lane * 4 < D) {
            v[c].x -= pivot; v[c].y -= pivot; v[c].z -= pivot; v[c].w -= pivot;
            s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
            q += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    const float mean = s / (float)D;                       // of the shifted row
    const float rs = rsqrtf(fmaxf(q / (float)D - mean * mean, 0.f) + 1e-6f);
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c * 256 + lane * 4 < D) { v[c].x -= mean; v[c].y -= mean; v[c].z -= mean; v[c].w -= mean; }
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int d = c * 256 + lane * 4;
        if (d < D) {
            // the reference runs the Linear under bf16 autocast: its input and weight are rounded to bf16
            const float y[4] = {bf16_to_f32(f32_to_bf16(v[c].x * rs * (1.0f + sc[c].x) + sh[c].x)),
                                bf16_to_f32(f32_to_bf16(v[c].y * rs * (1.0f + sc[c].y) + sh[c].y)),
                                bf16_to_f32(f32_to_bf16(v[c].z * rs * (1.0f + sc[c].z) + sh[c].z)),
                                bf16_to_f32(f32_to_bf16(v[c].w * rs * (1.0f + sc[c].w) + sh[c].w))};
#pragma unroll
            for (int o = 0; o < 16; ++o)
                if (o < a.Cout) {
                    if (wl) {      // (round 6) the weight rows, already rounded to bf16, from the workgroup's LDS image
                        const float4 w4 = *reinterpret_cast<const float4 *>(wl + (size_t)o * D + d);
                        acc[o] += y[0] * w4.x + y[1] * w4.y + y[2] * w4.z + y[3] * w4.w;
                    } else {
                        const float4 w4 = *reinterpret_cast<const float4 *>(a.w + (size_t)o * D + d);
                        acc[o] += y[0] * bf16_to_f32(f32_to_bf16(w4.x)) + y[1] * bf16_to_f32(f32_to_bf16(w4.y)) +
                                  y[2] * bf16_to_f32(f32_to_bf16(w4.z)) + y[3] * bf16_to_f32(f32_to_bf16(w4.w));
                    }
                }
        }
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) res[o] = o < a.Cout ? wave_sum(acc[o]) + a.bias[o] : 0.f;
}

// Round 6: the Cout x D weight (fp32, rounded to bf16 as the autocast Linear does) is staged once per workgroup into LDS when it fits the
// launch's dynamic allocation: every wave read all of it from the L2 before, Cout dependent round trips per 256 columns -- 26 us for the
// 768 rows x 10 channels of a stage-2 evaluation.  Same products, same order: same bits.
__global__ __launch_bounds__(256) void final_layer_kernel(FinalArgs a, int w_in_lds)
{
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    const float *wl = nullptr;
    if (w_in_lds) {   // kernel-uniform
        const int n4 = a.Cout * a.D / 4;
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 w4 = reinterpret_cast<const float4 *>(a.w)[i];
            reinterpret_cast<float4 *>(wlds)[i] = make_float4(bf16_to_f32(f32_to_bf16(w4.x)), bf16_to_f32(f32_to_bf16(w4.y)),
                                                             bf16_to_f32(f32_to_bf16(w4.z)), bf16_to_f32(f32_to_bf16(w4.w)));
        }
        __syncthreads();
        wl = wlds;
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (a.state == nullptr) {
        if (row >= a.M) return;
        float r[16];
        final_layer_row(a, row, lane, r, wl);
        if (lane < a.Cout) {
            float mine = 0.f;
#pragma unroll
            for (int o = 0; o < 16; ++o) mine = lane == o ? r[o] : mine;
            a.out[(size_t)row * a.Cout + lane] = mine;
        }
        return;
    }
    // fused sampler step.  Without CFG a wave owns one row; with CFG a workgroup owns two rows of the conditional half: waves 0, 1
    // evaluate them, waves 2, 3 their unconditional twins (round 5: one wave did both rows one after the other), combined through LDS
    const int half = a.cfg ? a.M / 2 : a.M;
    __shared__ float twin[2][16];
    float v[16];
    if (a.cfg) {
        const int w = threadIdx.x >> 6, pair = w & 1, role = w >> 1, r2 = blockIdx.x * 2 + pair;
        const bool have = r2 < half;
        if (have) final_layer_row(a, r2 + role * half, lane, v, wl);
        if (role == 1 && have && lane < 16) {
            float mine = 0.f;
#pragma unroll
            for (int o = 0; o < 16; ++o) mine = lane == o ? v[o] : mine;
            twin[pair][lane] = mine;
        }
        __syncthreads();
        if (role == 1 || !have) return;
#pragma unroll
        for (int o = 0; o < 16; ++o) v[o] = cfg_combine(v[o], twin[pair][o], a.cfg_scale);
    } else {
        if (row >= half) return;
        final_layer_row(a, row, lane, v, wl);
    }
    const int rowc = a.cfg ? blockIdx.x * 2 + ((threadIdx.x >> 6) & 1) : row;   // my row of the (conditional) half
    if (lane < a.Cout) {
        float mine = 0.f;
#pragma unroll
        for (int o = 0; o < 16; ++o) mine = lane == o ? v[o] : mine;
        if (a.dt == nullptr) {    // GaDitSamplerStep.velocity: the velocity itself, both halves
            for (int h = 0; h < (a.cfg ? 2 : 1); ++h) a.state[(size_t)(rowc + h * half) * a.Cout + lane] = mine;
            return;
        }
        const float dtv = *a.dt;
        float *slice = a.traj ? a.traj + (size_t)(*a.counter + 1) * a.traj_stride : nullptr;
        for (int h = 0; h < (a.cfg ? 2 : 1); ++h) {
            const size_t i = (size_t)(rowc + h * half) * a.Cout + lane;
            const float y = euler_update(a.state[i], dtv, mine);   // both halves hold the same state and receive the same update
            a.state[i] = y;
            if (slice) slice[i] = y;
        }
    }
}

__global__ void sampler_advance_kernel(int *counter, const float *t_grid, const float *dt_grid, int grid_len, float *timesteps,
                                       int batch, float *dt)
{
    const int c = min(*counter + 1, grid_len - 1);
    __syncthreads();
    if ((int)threadIdx.x < batch) timesteps[threadIdx.x] = t_grid[c];
    if (threadIdx.x == 0) { *dt = dt_grid[c]; *counter = *counter + 1; }
}

}  // namespace gadit

extern "C" int ga_rmsnorm_modulate(const GaRmsNormArgs *a, void *stream)
{
    using namespace gadit;
    if (!a || !a->x || !a->weight || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->M <= 0 || a->D <= 0 || a->D % 4 != 0 || a->D > 2048 || a->rows_per_batch <= 0 ||
        ((a->scale == nullptr) != (a->shift == nullptr)))
        return GA_DIT_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(rmsnorm_modulate_kernel, dim3((a->M + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}

extern "C" int ga_small_linear(const GaSmallLinearArgs *a, void *stream)
{
    using namespace gadit;
    if (!a || !a->x || !a->W || !a->y) return GA_DIT_ERR_NULL_ARG;
    if (a->B <= 0 || a->B > 16 || a->N <= 0 || a->K <= 0 || a->K % 8 != 0 || (a->act_in == 2 && a->K != 256)) return GA_DIT_ERR_BAD_SHAPE;
    gadit::ModOut none{};
    hipLaunchKernelGGL(small_linear_kernel, dim3((a->N + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *a, none);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}

extern "C" int ga_dit_shift_bias(const ga_bf16 *W, int32_t w_tiled, const float *bias, int32_t N, int32_t K, const float *shift,
                                 int64_t shift_stride, int32_t batch, float *out, void *stream)
{
    using namespace gadit;
    if (!W || !shift || !out) return GA_DIT_ERR_NULL_ARG;
    if (N <= 0 || N % 8 != 0 || K <= 0 || K % 64 != 0 || K > 4096 || batch <= 0 || shift_stride % 4 != 0) return GA_DIT_ERR_BAD_SHAPE;
    ShiftBiasArgs sb{};
    sb.W[0] = W; sb.bias[0] = bias; sb.shift = shift; sb.out = out; sb.shift_batch_stride = shift_stride;
    sb.N0 = N; sb.N1 = 0; sb.K = K; sb.B = batch; sb.jobs = 1; sb.tiled = w_tiled ? 1 : 0;
    hipLaunchKernelGGL(shift_bias_kernel, dim3((unsigned)shift_bias_wgs(N, 0)), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), sb);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}

// =================================================================================================================
// Whole forward: launch sequence of one function evaluation (see include/ga_dit.h).  ~11 launches per block; nothing
// synchronises, so the caller can capture it in a HIP graph.
namespace gadit {

struct Ws {
    float *xres, *tfreq, *t1, *pln, *pvec, *tvec, *t0, *mod, *rowss, *sbias;
    uint16_t *xn, *qkv, *att, *hmid, *vt;
    size_t vt_bytes;
    void *splitk;          // scratch of the deterministic split-K of fc2 (GaGemmArgs.splitk_ws); nullptr above 3072 rows
    size_t splitk_bytes;
    size_t total;
};

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static Ws carve(const GaDitModel *m, int B, int L, void *base)
{
    const size_t M = (size_t)B * L, D = m->hidden;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return o; };
    unsigned char *p = static_cast<unsigned char *>(base);
    Ws w;
    const size_t Lp = ((size_t)L + 63) / 64 * 64;
    const size_t qkv_cols = 2;      // q | k row-major; v goes to the V^T image (round 6: for every head dim)
    const size_t o_xres = take(M * D * 4), o_xn = take(M * D * 2), o_qkv = take(M * qkv_cols * D * 2), o_att = take(M * D * 2);
    const size_t o_vt = take((size_t)B * D * Lp * 2);
    const size_t o_hmid = take(M * 4 * D * 2), o_tfreq = take((size_t)B * 256 * 4), o_t1 = take((size_t)B * D * 4);
    const size_t o_pln = take((size_t)B * m->context_dim * 4), o_pvec = take((size_t)B * D * 4);
    const size_t o_tvec = take((size_t)B * D * 4), o_t0 = take((size_t)B * 6 * D * 4);
    const size_t o_mod = take((size_t)m->depth * B * 6 * D * 4);
    const size_t o_rowss = take(M * (((D / 64) + 3) & ~(size_t)3) * 4);   // per-row partial sums of squares of the residual stream (folded pre-norm)
    const size_t o_sbias = take((size_t)m->depth * B * 7 * D * 4);   // shift_b W^T + bias of the qkv and fc1 projections (folded modulated pre-norms)
    // (round 6) split-K scratch of the MLP's second linear: counters + partial tiles; only where a 4-way split can fill the chip
    // (fc2: up to 4 splits of [M, D]; qkv / fc1 at 768 rows: 2 splits of [M, 4 D] -- half of the 4-split bound)
    // Measured slower (include/ga_dit.h: ga_gemm_splitk_mode) -- the scratch only exists when GA_GEMM_SPLITK asks for a split at process start
    static const bool sk_on = [] { const char *e = getenv("GA_GEMM_SPLITK"); return e && atoi(e) > 0; }();
    w.splitk_bytes = sk_on && M <= 3072 ? std::max(ga_gemm_splitk_workspace_bytes((int32_t)M, (int32_t)D),
                                          GA_GEMM_SPLITK_COUNTER_BYTES + (ga_gemm_splitk_workspace_bytes((int32_t)M, (int32_t)(4 * D)) - GA_GEMM_SPLITK_COUNTER_BYTES) / 2) : 0;
    const size_t o_sk = take(w.splitk_bytes);
    w.splitk = w.splitk_bytes && base ? p + o_sk : nullptr;
    w.total = off;
    w.xres = reinterpret_cast<float *>(p + o_xres); w.xn = reinterpret_cast<uint16_t *>(p + o_xn);
    w.qkv = reinterpret_cast<uint16_t *>(p + o_qkv); w.att = reinterpret_cast<uint16_t *>(p + o_att);
    w.vt = reinterpret_cast<uint16_t *>(p + o_vt); w.vt_bytes = (size_t)B * D * Lp * 2;
    w.hmid = reinterpret_cast<uint16_t *>(p + o_hmid); w.tfreq = reinterpret_cast<float *>(p + o_tfreq);
    w.t1 = reinterpret_cast<float *>(p + o_t1); w.pln = reinterpret_cast<float *>(p + o_pln);
    w.pvec = reinterpret_cast<float *>(p + o_pvec); w.tvec = reinterpret_cast<float *>(p + o_tvec);
    w.t0 = reinterpret_cast<float *>(p + o_t0); w.mod = reinterpret_cast<float *>(p + o_mod);
    w.rowss = reinterpret_cast<float *>(p + o_rowss); w.sbias = reinterpret_cast<float *>(p + o_sbias);
    return w;
}

// block i's cross-attention pre-norm can be folded into fc2 of block i-1 and its own q projection (include/ga_dit.h)
static bool can_fold(const GaDitModel *m, int i)
{
    // heads of 64 (the fused q projection of the attention kernel reads the partial sums four at a time): widths the four-slot GEMM
    // instances serve; other head dims (round 6, DiT-PixArt-PCD-CLAY-XL): those and the three-slot widths, up to 20 partial sums per row
    const int D = m->hidden;
    const bool w256 = D % 256 == 0 && D <= 1024;
    const bool width_ok = m->hidden / m->heads == 64 ? w256 : (w256 || (D % 192 == 0 && D <= 1280));
    return m->blocks[i].ca_q_w_prenorm != nullptr && width_ok;
}

static bool model_ok(const GaDitModel *m)
{
    return m && m->blocks && m->hidden > 0 && m->hidden % 64 == 0 && m->hidden <= 2048 && m->depth > 0 && m->depth <= 64 &&
           m->heads > 0 && m->hidden % m->heads == 0 && (m->hidden / m->heads) % 8 == 0 && m->hidden / m->heads <= 128 && m->in_channels > 0 && m->in_channels <= 16 && m->out_channels > 0 &&
           m->out_channels <= 16 && m->context_dim > 0 && m->context_dim % 64 == 0;
}

}  // namespace gadit

extern "C" const char *ga_dit_version(void) { return "ga_mi355 dit gfx950 r6"; }

extern "C" size_t ga_dit_workspace_bytes(const GaDitModel *m, int32_t batch, int32_t tokens, int32_t ctx_tokens)
{
    (void)ctx_tokens;
    if (!gadit::model_ok(m) || batch <= 0 || tokens <= 0) return 0;
    return gadit::carve(m, batch, tokens, nullptr).total;
}

#define GA_TRY(expr) do { const int rc_ = (expr); if (rc_ != GA_DIT_OK) return rc_; } while (0)
// GA_DIT_ABLATE builds (tools/dit_ablate.sh) drop whole launch classes from the evaluation -- wrong results, wall-time shares only:
// env GA_DIT_SKIP bit mask 1 self-attention, 2 cross-attention, 4 RMSNorm launches, 8 fc1 + fc2, 16 qkv + proj, 32 CA q + out
#ifdef GA_DIT_ABLATE
static int ga_skip_mask() { static const int v = [] { const char *e = getenv("GA_DIT_SKIP"); return e ? atoi(e) : 0; }(); return v; }
#define GA_UNLESS(bit, expr) do { if (!(ga_skip_mask() & (bit))) GA_TRY(expr); } while (0)
#else
#define GA_UNLESS(bit, expr) GA_TRY(expr)
#endif

extern "C" int ga_dit_cache_context(const GaDitModel *m, int32_t batch, int32_t ctx_tokens, const ga_bf16 *ctx,
                                    ga_bf16 *ca_k, ga_bf16 *ca_vt, void *stream)
{
    if (!gadit::model_ok(m) || !ctx || !ca_k || !ca_vt) return GA_DIT_ERR_NULL_ARG;
    if (batch <= 0 || ctx_tokens <= 0) return GA_DIT_ERR_BAD_SHAPE;
    const int rows = batch * ctx_tokens, D = m->hidden;
    const int64_t Mp = ((int64_t)ctx_tokens + 63) / 64 * 64;
    const bool hd64 = D / m->heads == 64;   // other head dims (ga_attention_hd_bf16): the same K | V^T images, k's per-head norm as a pass of its own
    for (int i = 0; i < m->depth; ++i) {
        GaGemmArgs g{};
        g.M = rows; g.N = 2 * D; g.K = m->context_dim; g.epilogue = GA_GEMM_EPI_STORE_BF16;
        g.A = ctx; g.lda = m->context_dim; g.W = m->blocks[i].ca_kv_w; g.w_tiled = m->gemm_weights_tiled; g.bias = nullptr;
        g.out = ca_k + (size_t)i * rows * D; g.ldo = D;                    // K columns [0, D)
        g.vt = ca_vt + (size_t)i * batch * D * Mp; g.vt_col0 = D; g.vt_ld = Mp; g.rows_per_batch = ctx_tokens;
        if (hd64) { g.qk_w0 = m->blocks[i].ca_k_norm_w; g.qk_cols0 = D; g.qk_cols1 = D; }  // k_norm applied once, here
        GA_TRY(ga_gemm_bf16(&g, stream));
        if (!hd64) GA_TRY(ga_head_rmsnorm_bf16(ca_k + (size_t)i * rows * D, rows, D, m->heads, D / m->heads, m->blocks[i].ca_k_norm_w, stream));
    }
    return GA_DIT_OK;
}

namespace gadit {
// The conditioning chain of an evaluation (timestep embedding -> + pooled vector -> adaLN -> modulation tables [-> block 0's shift
// rows]) is three or four small dependent launches, ~30 us of pure latency, and nothing in front of block 0's cross-attention needs its
// results: it CAN run on a helper stream beside the token embedding, its GEMM and block 0's pre-norm, forked from / joined to the caller's
// stream with events (capture-safe: the helper stream joins a capture of the caller's stream and leaves it at the join) -- built and
// measured in round 5, slower (see ga_dit_forward), kept behind GA_DIT_FORK=1.
struct SideStream {
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool ok = false;
};
static SideStream &side_stream()
{
    static thread_local SideStream ss = [] {
        SideStream f;
        f.ok = hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&f.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&f.join, hipEventDisableTiming) == hipSuccess;
        return f;
    }();
    return ss;
}
}  // namespace gadit

extern "C" int ga_dit_pooled_vector(const GaDitModel *m, int32_t batch, const float *img_vector, float *scratch, float *out, void *stream)
{
    using namespace gadit;
    if (!model_ok(m) || !img_vector || !scratch || !out) return GA_DIT_ERR_NULL_ARG;
    if (batch <= 0 || batch > 16) return GA_DIT_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(layernorm_rows_kernel, dim3(batch), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), img_vector, m->pool_ln_w,
                       m->pool_ln_b, scratch, m->context_dim, 1e-5f);
    GaSmallLinearArgs l2{batch, m->hidden, m->context_dim, 0, 0, scratch, m->pool_w, m->pool_b, nullptr, out};
    return ga_small_linear(&l2, stream);
}

extern "C" int ga_dit_forward(const GaDitModel *m, const GaDitForwardArgs *a, void *stream)
{
    using namespace gadit;
    if (!model_ok(m) || !a) return GA_DIT_ERR_NULL_ARG;
    if (!a->x || !a->timesteps || !a->img_vector || !a->ca_k || !a->ca_vt || (!a->out && !a->step) || !a->workspace) return GA_DIT_ERR_NULL_ARG;
    if (m->stage2 && !a->fps_xyz) return GA_DIT_ERR_NULL_ARG;
    const int B = a->batch, L = a->tokens, D = m->hidden, Mrows = B * L, hd = D / m->heads;
    if (B <= 0 || B > 16 || L <= 0 || a->ctx_tokens <= 0) return GA_DIT_ERR_BAD_SHAPE;
    // a sampler step writes states / velocities of in_channels floats per token: a model that also predicts sigma (out_channels !=
    // in_channels) has no such step -- refused before anything is enqueued (the reference trips body_fn's shape assert there)
    if (a->step && (m->out_channels != m->in_channels || (a->step->cfg && (B % 2 != 0)))) return GA_DIT_ERR_BAD_SHAPE;
    const Ws w = carve(m, B, L, a->workspace);
    if (a->workspace_bytes < w.total || ((uintptr_t)a->workspace & 255)) return GA_DIT_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    // the tile counters of the split-K GEMMs are left zero by every launch; cleared once per evaluation all the same, so that a
    // workspace needs no initialisation and an aborted launch cannot poison the next evaluation (a 16 KiB memset node)
    if (w.splitk && hipMemsetAsync(w.splitk, 0, GA_GEMM_SPLITK_COUNTER_BYTES, s) != hipSuccess) return GA_DIT_ERR_LAUNCH;

    // ---- conditioning path: t = t_embedder(timesteps) + pooled_vec_embedder(img_vector); t0 = adaLN(SiLU(t))
    // (round 5: the sinusoidal features are formed inside the first linear -- one launch less -- and the pooled-vector branch, which
    //  does not depend on the time, is taken from the caller when it has been computed once per conditioning: ga_dit_pooled_vector)
    // MEASURED SLOWER, off by default (GA_DIT_FORK=1 switches it on): DiT-L 3.05 -> 3.14 ms per step in the replayed Euler graph, 3.09 -> 3.20 per
    // dopri5 evaluation, 3.06 -> 3.16 eager -- a cross-stream dependency costs this runtime ~45 us per hop, more than the ~25 us of
    // chain it hides (the same sign as round 4's two concurrent CFG halves)
    static const bool fork_env = [] { const char *e = getenv("GA_DIT_FORK"); return e && atoi(e) != 0; }();
    SideStream *fk = fork_env ? &side_stream() : nullptr;
    if (fk && (!fk->ok || hipEventRecord(fk->fork, s) != hipSuccess || hipStreamWaitEvent(fk->side, fk->fork, 0) != hipSuccess)) fk = nullptr;
    hipStream_t cs = fk ? fk->side : s;      // the conditioning chain's stream
    void *cstream = reinterpret_cast<void *>(cs);
    bool joined = fk == nullptr;
    auto join = [&]() {   // the caller's stream waits for the chain (before its first consumer: block 0's cross-attention launch and its tail)
        if (joined) return true;
        joined = true;
        return hipEventRecord(fk->join, fk->side) == hipSuccess && hipStreamWaitEvent(s, fk->join, 0) == hipSuccess;
    };
    GaSmallLinearArgs l1{B, D, 256, 2, 1, a->timesteps, m->t_mlp0_w, m->t_mlp0_b, nullptr, w.t1};
    if (ga_small_linear(&l1, cstream) != GA_DIT_OK) { join(); return GA_DIT_ERR_LAUNCH; }
    const float *pvec = a->pooled_vec;
    if (!pvec) {
        if (ga_dit_pooled_vector(m, B, a->img_vector, w.pln, w.pvec, cstream) != GA_DIT_OK) { join(); return GA_DIT_ERR_LAUNCH; }
        pvec = w.pvec;
    }
    GaSmallLinearArgs l3{B, D, D, 0, 0, w.t1, m->t_mlp2_w, m->t_mlp2_b, pvec, w.tvec};
    if (ga_small_linear(&l3, cstream) != GA_DIT_OK) { join(); return GA_DIT_ERR_LAUNCH; }
    {   // t0 = adaLN(SiLU(t)), and with it mod[blk][b] = scale_shift_table_blk + t0[b] of every block (dit_models_xformers.py:769-770)
        GaSmallLinearArgs l4{B, 6 * D, D, 1, 0, w.tvec, m->adaln_w, m->adaln_b, nullptr, w.t0};
        ModOut mo{};
        for (int i = 0; i < m->depth; ++i) mo.tables[i] = m->blocks[i].scale_shift_table;
        mo.mod = w.mod; mo.depth = m->depth;
        hipLaunchKernelGGL(small_linear_kernel, dim3((l4.N + 3) / 4), dim3(256), 0, cs, l4, mo);
    }
    // the modulated pre-norms of the self-attention and the MLP fold into the neighbouring GEMMs like the cross-attention's (below):
    // their shifts go through the qkv / fc1 weights once per evaluation, for all blocks in one launch
    // (measured, DiT-L: 1536 rows 3.42 -> 3.22 ms per evaluation, DiT-B 1.49 -> 1.40; at 6144 rows the stand-alone norm launches are
    //  bandwidth-sized and the fold is 1 % behind -- left off there)
    static const bool fold_mod_env = [] { const char *e = getenv("GA_DIT_FOLD_MOD"); return !e || atoi(e) != 0; }();   // A/B aid
    static const int fold_rows = [] { const char *e = getenv("GA_DIT_FOLD_ROWS"); return e ? atoi(e) : 3072; }();                      // A/B aid
    const bool fold_mod = fold_mod_env && can_fold(m, 0) && m->depth <= 64 && Mrows <= fold_rows;
    static const bool sb_tail_env = [] { const char *e = getenv("GA_DIT_SBTAIL"); return !e || atoi(e) != 0; }();
    // the shift rows of block i + 1 ride behind the self-attention grid of block i while that grid leaves CUs idle (a CFG pair: 192
    // workgroups + 56 of the tail on 256 CUs); on a full grid they would queue behind it (8 items: 9.2 -> 9.8 ms) -- one launch up front then
    // compute units of the device the launches go to (the tails only ride where the attention grid leaves CUs idle): 256 on a whole MI355X,
    // fewer in a partitioned mode; asked once per process
    static const int ncu = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; return n; }();
    bool sb_tail = false, sb_tail0 = false;
    const int ca_batch = (a->ca_batch <= 0 || a->ca_batch > B) ? B : a->ca_batch;
    if (fold_mod && sb_tail_env && hd != 64) {
        // other head dims (round 6): block i + 1's rows behind the SELF-attention grid of block i where that grid leaves the CUs free and its
        // workgroups have the waves for the job (block 0's: the launch up front); none behind the cross-attention grid, which carries the prefetch
        const GaAttentionHdArgs probe{B, m->heads, L, L, hd, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, 0, nullptr, 0, nullptr, nullptr};
        sb_tail = attention_hd_workgroups(&probe) + shift_bias_wgs(3 * D, 4 * D) <= ncu && attention_hd_hosts_shift_bias(&probe, D);
    }
    if (fold_mod && sb_tail_env && hd == 64) {
        const GaAttentionArgs probe{B, m->heads, L, L, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, 0};
        sb_tail = attention_workgroups(&probe) + shift_bias_wgs(3 * D, 4 * D) <= ncu;
        // round 5: block 0's rows ride the same way behind block 0's CROSS-attention grid (its qkv projection is the first consumer):
        // the stand-alone 12 us launch in front of the blocks is gone when that grid leaves the CUs free as well
        const GaAttentionArgs probe_ca{ca_batch, m->heads, L, a->ctx_tokens, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, 0};
        static const bool sb_tail0_env = [] { const char *e = getenv("GA_DIT_SBTAIL0"); return !e || atoi(e) != 0; }();   // A/B aid
        sb_tail0 = sb_tail0_env && sb_tail && attention_workgroups(&probe_ca) + shift_bias_wgs(3 * D, 4 * D) <= ncu;
    }
    // Round 6: WEIGHT PREFETCH.  A GEMM whose weights sit in the Infinity Cache instead of HBM starts and streams faster (fc2 at 1536
    // rows 24.0 -> 21.1 us, at 768 rows 19.3 -> 15.7; fc1 18.0 -> 17.1; tools/warm_vs_cold.py), and every attention grid of a CFG pair
    // leaves 64 of the 256 CUs idle.  The tail workgroups behind block i's CROSS-attention grid (dit_attention.hip: PrefetchJob) read
    // block i's fc2 and self-attention output weights and block i + 1's cross-attention q / output weights -- plain loads nobody
    // waits for; the qkv and fc1 weights are already passed over by the shift rows' tail a block earlier.  Same-box A/B
    // (profiles/r6_prefetch_ab.txt): DiT-L 2.91 -> 2.73 ms per evaluation (-6 %), batch 1 2.53 -> 2.38, CFG batch 4 4.79 -> 4.67, DiT-B
    // 1.29 -> 1.27.  GA_DIT_PREFETCH: 0 off | 1 fc2 only | 2 fc2 behind the self-attention grid instead | 3 fc2 + proj | 4 (default) 3 + the
    // next block's cross-attention q / out | 5: 4 + its cached K / V^T (no gain) | 6: 4 + this block's qkv / fc1 once more (no gain)
    static const int pf_mode = [] { const char *e = getenv("GA_DIT_PREFETCH"); return e ? atoi(e) : 4; }();
    int pf_ca = 0, pf_sa = 0;        // tail workgroups available for it
    if (hd == 64 && pf_mode > 0) {
        const GaAttentionArgs probe_sa{B, m->heads, L, L, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, 0};
        const GaAttentionArgs probe_ca{ca_batch, m->heads, L, a->ctx_tokens, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, 0};
        pf_ca = std::min(64, std::max(0, ncu - attention_workgroups(&probe_ca)));
        pf_sa = std::min(64, std::max(0, ncu - attention_workgroups(&probe_sa)));
        if (pf_ca < 16 || pf_mode == 2) pf_ca = 0;
        if (pf_sa < 16 || pf_mode != 2) pf_sa = 0;
    }
    if (fold_mod && !sb_tail0) {
        ShiftBiasArgs sb{};
        for (int i = 0; i < m->depth; ++i) {
            sb.W[2 * i] = m->blocks[i].qkv_w; sb.bias[2 * i] = m->blocks[i].qkv_b;
            sb.W[2 * i + 1] = m->blocks[i].fc1_w; sb.bias[2 * i + 1] = m->blocks[i].fc1_b;
        }
        sb.shift = w.mod; sb.out = w.sbias; sb.shift_block_stride = (long long)B * 6 * D; sb.shift_batch_stride = 6 * (long long)D;
        sb.shift_which_off = 3 * (long long)D; sb.out_block_stride = (long long)B * 7 * D;
        // (GA_DIT_SBTAIL=0: all blocks here, A/B aid; default: block 0 here, block i + 1 behind the self-attention grid of block i)
        sb.N0 = 3 * D; sb.N1 = 4 * D; sb.K = D; sb.B = B; sb.jobs = sb_tail ? 1 : m->depth; sb.tiled = m->gemm_weights_tiled;
        hipLaunchKernelGGL(shift_bias_kernel, dim3((unsigned)(sb.jobs * shift_bias_wgs(sb.N0, sb.N1))), dim3(1024), 0, cs, sb);
    }
    // every early return below joins first (a forked capture must be rejoined)
#undef GA_TRY
#define GA_TRY(expr) do { const int rc_ = (expr); if (rc_ != GA_DIT_OK) { join(); return rc_; } } while (0)
    // ---- token embedding: x = fc2(gelu_tanh(fc1(x))) (+ xyz positional embedding)
    {
        EmbedArgs e{Mrows, D, m->in_channels, m->stage2, a->x, m->xe_fc1_w, m->xe_fc1_b, a->fps_xyz, m->xyz_w, m->xyz_b,
                    w.xn, w.xres};
        hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((Mrows + kET - 1) / kET), (unsigned)((D + 127) / 128)), dim3(256), 0, s, e);
        GaGemmArgs g{};
        g.M = Mrows; g.N = D; g.K = D; g.epilogue = GA_GEMM_EPI_RESIDUAL; g.A = w.xn; g.lda = D; g.W = m->xe_fc2_w; g.w_tiled = m->gemm_weights_tiled;
        g.bias = m->xe_fc2_b; g.out = w.xres; g.ldo = D; g.gate = nullptr; g.rows_per_batch = L;
        GA_TRY(ga_gemm_bf16(&g, stream));
    }
    const size_t kv_rows = (size_t)B * a->ctx_tokens;
    const int64_t Mp = ((int64_t)a->ctx_tokens + 63) / 64 * 64, Lp = ((int64_t)L + 63) / 64 * 64;
    if (Lp != L && hipMemsetAsync(w.vt, 0, w.vt_bytes, s) != hipSuccess) { join(); return GA_DIT_ERR_LAUNCH; }  // zero key padding
    const int Mca = ca_batch * L;
    for (int i = 0; i < m->depth; ++i) {
        const GaDitBlockWeights &bw = m->blocks[i];
        const float *mod = w.mod + (size_t)i * B * 6 * D;  // [B][6][D]: shift_msa scale_msa gate_msa shift_mlp scale_mlp gate_mlp
        // cross-attention on the image tokens.  Its pre-norm has no modulation, so from the second block on it can be folded
        // into the neighbouring GEMMs: the previous block's fc2 epilogue left bf16(x) in xn and the rows' sums of squares in
        // rowss (below); the q projection, with the norm weight folded into its columns, applies rsqrt(mean + eps) to its rows.
        const bool folded = i > 0 && can_fold(m, i);
        if (hd != 64) {
            // HEAD DIMS OTHER THAN 64 (DiT-PixArt-PCD-CLAY-XL: 16 heads of 72): the same block without the attention-launch tails -- k's per-head
            // norm as a pass of its own (the GEMM epilogue's is 64-wide), ga_attention_hd_bf16 on q | k row-major + V^T
            // round 6: the three pre-norms folded into the neighbouring GEMMs as on the 64-wide path (fold_mod; `folded`: this block's
            // cross-attention pre-norm, whose operand the previous block's fc2 left behind) -- three launches per block less
            const float *sbias_x = w.sbias + (size_t)i * B * 7 * D;   // [B][3D] qkv | [B][4D] fc1
            if (!(fold_mod && folded)) {
                GaRmsNormArgs n0{Mca, D, L, w.xres, bw.prenorm_ca_w, nullptr, nullptr, 0, w.xn, nullptr, 0};
                GA_TRY(ga_rmsnorm_modulate(&n0, stream));
            }
            GaGemmArgs gq{};
            gq.M = Mca; gq.N = D; gq.K = D; gq.epilogue = GA_GEMM_EPI_STORE_BF16; gq.A = w.xn; gq.lda = D;
            gq.W = (fold_mod && folded) ? bw.ca_q_w_prenorm : bw.ca_q_w;
            gq.w_tiled = m->gemm_weights_tiled; gq.out = w.qkv; gq.ldo = D;
            if (fold_mod && folded) { gq.row_ss = w.rowss; gq.row_ss_tiles = D / 64; gq.row_ss_dim = D; gq.row_ss_eps = 1e-5f; }
            GA_TRY(ga_gemm_bf16(&gq, stream));
            if (i == 0 && !join()) return GA_DIT_ERR_LAUNCH;
            // (V^T of the image tokens from the cache like the 64-wide path, q's norm inside the attention kernel)
            GaAttentionHdArgs ca{ca_batch, m->heads, L, a->ctx_tokens, hd, w.qkv, a->ca_k + (size_t)i * kv_rows * D, nullptr, D, D, 0, w.att, D,
                                 a->ca_vt + (size_t)i * B * D * Mp, Mp, bw.ca_q_norm_w, nullptr};
            // the weight prefetch of the 64-wide path (GA_DIT_PREFETCH) behind this grid as well: this block's fc2 and self-attention output
            // weights, the next block's cross-attention q / output weights, by the CUs the grid leaves idle
            PrefetchJob pfx{};
            int pfx_wgs = 0;
            if (pf_mode > 0) {
                const unsigned DD = (unsigned)D * (unsigned)D;
                pfx.ptr[0] = reinterpret_cast<const char *>(bw.fc2_w); pfx.bytes[0] = (8u * DD) / 1024u * 1024u;
                if (pf_mode >= 3) { pfx.ptr[1] = reinterpret_cast<const char *>(bw.proj_w); pfx.bytes[1] = (2u * DD) / 1024u * 1024u; }
                if (pf_mode >= 4) {
                    const int ni = i + 1 < m->depth ? i + 1 : 0;
                    const GaDitBlockWeights &nb = m->blocks[ni];
                    pfx.ptr[2] = reinterpret_cast<const char *>(nb.ca_out_w); pfx.bytes[2] = (2u * DD) / 1024u * 1024u;
                    const uint16_t *qw = (fold_mod && ni > 0 && can_fold(m, ni) && nb.ca_q_w_prenorm) ? nb.ca_q_w_prenorm : nb.ca_q_w;
                    pfx.ptr[3] = reinterpret_cast<const char *>(qw); pfx.bytes[3] = (2u * DD) / 1024u * 1024u;
                }
                pfx_wgs = std::min(64, std::max(0, ncu - attention_hd_workgroups(&ca)));
                if (pfx_wgs < 16) pfx_wgs = 0;
            }
            GA_TRY(attention_hd_with_tail(&ca, stream, pfx_wgs ? &pfx : nullptr, pfx_wgs));
            GaGemmArgs go{};
            go.M = Mca; go.N = D; go.K = D; go.epilogue = GA_GEMM_EPI_RESIDUAL; go.A = w.att; go.lda = D; go.W = bw.ca_out_w; go.w_tiled = m->gemm_weights_tiled;
            go.bias = bw.ca_out_b; go.out = w.xres; go.ldo = D; go.gate = nullptr; go.rows_per_batch = L;
            if (fold_mod) {     // over ALL rows (the items that skipped the cross-attention take the bias with a zero product), emitting norm1's operand
                go.M = Mrows; go.k_rows = Mca;
                go.emit_x = w.xn; go.emit_ld = D; go.emit_ss = w.rowss; go.emit_w = bw.norm1_w; go.emit_scale = mod + 1 * D; go.emit_scale_stride = 6 * (int64_t)D;
            }
            GA_TRY(ga_gemm_bf16(&go, stream));
            if (!fold_mod) {
                GaRmsNormArgs n1{Mrows, D, L, w.xres, bw.norm1_w, mod + 1 * D, mod + 0 * D, 6 * (int64_t)D, w.xn, Mca < Mrows ? bw.ca_out_b : nullptr, Mca};
                GA_TRY(ga_rmsnorm_modulate(&n1, stream));
            }
            GaGemmArgs gqkv{};
            gqkv.M = Mrows; gqkv.N = 3 * D; gqkv.K = D; gqkv.epilogue = GA_GEMM_EPI_STORE_BF16; gqkv.A = w.xn; gqkv.lda = D; gqkv.W = bw.qkv_w;
            gqkv.w_tiled = m->gemm_weights_tiled; gqkv.bias = bw.qkv_b; gqkv.out = w.qkv; gqkv.ldo = 2 * D;      // q | k row-major ...
            gqkv.vt = w.vt; gqkv.vt_col0 = 2 * D; gqkv.vt_ld = Lp; gqkv.rows_per_batch = L;                      // ... v transposed (width-generic)
            if (fold_mod) {
                gqkv.row_ss = w.rowss; gqkv.row_ss_tiles = D / 64; gqkv.row_ss_dim = D; gqkv.row_ss_eps = 1e-5f;
                gqkv.bias = sbias_x; gqkv.bias_stride = 3 * (int64_t)D;
            }
            GA_TRY(ga_gemm_bf16(&gqkv, stream));
            // (k's per-head norm inside the attention kernel too, GA_DIT_HD_KNORM=0: as a launch of its own -- A/B aid)
            static const bool knorm_in = [] { const char *e = getenv("GA_DIT_HD_KNORM"); return !e || atoi(e) != 0; }();
            if (!knorm_in) GA_TRY(ga_head_rmsnorm_bf16(w.qkv + D, Mrows, 2 * D, m->heads, hd, bw.k_norm_w, stream));
            GaAttentionHdArgs sa{B, m->heads, L, L, hd, w.qkv, w.qkv + D, nullptr, 2 * D, 2 * D, 0, w.att, D, w.vt, Lp, bw.q_norm_w, knorm_in ? bw.k_norm_w : nullptr};
            if (fold_mod && sb_tail && i + 1 < m->depth) {
                const GaDitBlockWeights &nb = m->blocks[i + 1];
                ShiftBiasJob job{{nb.qkv_w, nb.fc1_w}, {nb.qkv_b, nb.fc1_b}, w.mod + (size_t)(i + 1) * B * 6 * D, w.sbias + (size_t)(i + 1) * B * 7 * D,
                                 6 * (long long)D, 3 * (long long)D, 3 * D, 4 * D, D, B, m->gemm_weights_tiled};
                GA_TRY(attention_hd_with_tail(&sa, stream, nullptr, 0, &job));
            } else
                GA_TRY(ga_attention_hd_bf16(&sa, stream));
            GaGemmArgs gp{};
            gp.M = Mrows; gp.N = D; gp.K = D; gp.epilogue = GA_GEMM_EPI_RESIDUAL; gp.A = w.att; gp.lda = D; gp.W = bw.proj_w; gp.w_tiled = m->gemm_weights_tiled;
            gp.bias = bw.proj_b; gp.out = w.xres; gp.ldo = D; gp.gate = mod + 2 * D; gp.gate_stride = 6 * (int64_t)D; gp.rows_per_batch = L;
            if (fold_mod) {
                gp.emit_x = w.xn; gp.emit_ld = D; gp.emit_ss = w.rowss; gp.emit_w = bw.norm2_w; gp.emit_scale = mod + 4 * D; gp.emit_scale_stride = 6 * (int64_t)D;
            }
            GA_TRY(ga_gemm_bf16(&gp, stream));
            if (!fold_mod) {
                GaRmsNormArgs n2{Mrows, D, L, w.xres, bw.norm2_w, mod + 4 * D, mod + 3 * D, 6 * (int64_t)D, w.xn, nullptr, 0};
                GA_TRY(ga_rmsnorm_modulate(&n2, stream));
            }
            GaGemmArgs g1{};
            g1.M = Mrows; g1.N = 4 * D; g1.K = D; g1.epilogue = GA_GEMM_EPI_GELU_BF16; g1.A = w.xn; g1.lda = D; g1.W = bw.fc1_w; g1.w_tiled = m->gemm_weights_tiled;
            g1.bias = bw.fc1_b; g1.out = w.hmid; g1.ldo = 4 * D;
            if (fold_mod) {
                g1.row_ss = w.rowss; g1.row_ss_tiles = D / 64; g1.row_ss_dim = D; g1.row_ss_eps = 1e-5f;
                g1.bias = sbias_x + (size_t)B * 3 * D; g1.bias_stride = 4 * (int64_t)D; g1.rows_per_batch = L;
            }
            GA_TRY(ga_gemm_bf16(&g1, stream));
            GaGemmArgs g2{};
            g2.M = Mrows; g2.N = D; g2.K = 4 * D; g2.epilogue = GA_GEMM_EPI_RESIDUAL; g2.A = w.hmid; g2.lda = 4 * D; g2.W = bw.fc2_w;
            g2.w_tiled = m->gemm_weights_tiled; g2.bias = bw.fc2_b; g2.out = w.xres; g2.ldo = D; g2.gate = mod + 5 * D; g2.gate_stride = 6 * (int64_t)D;
            g2.rows_per_batch = L;
            if (fold_mod && i + 1 < m->depth && can_fold(m, i + 1)) { g2.emit_x = w.xn; g2.emit_ld = D; g2.emit_ss = w.rowss; }
            GA_TRY(ga_gemm_bf16(&g2, stream));
            continue;
        }
        if (!folded) {
            GaRmsNormArgs n0{Mca, D, L, w.xres, bw.prenorm_ca_w, nullptr, nullptr, 0, w.xn, nullptr, 0};
            GA_UNLESS(4, ga_rmsnorm_modulate(&n0, stream));
        }
        GaAttentionArgs ca{ca_batch, m->heads, L, a->ctx_tokens, w.qkv, a->ca_k + (size_t)i * kv_rows * D,
                           a->ca_vt + (size_t)i * B * D * Mp, D, D, Mp, nullptr, nullptr, w.att, D};
        // round 5: the q projection inside the attention workgroups when the cross-attention runs its 64-query configuration (one
        // sample's conditional half): one launch per block less, q never goes through memory (GA_DIT_FUSE_Q=0: the GEMM launch, A/B aid)
        static const bool fuse_q_env = [] { const char *e = getenv("GA_DIT_FUSE_Q"); return !e || atoi(e) != 0; }();
        if (fuse_q_env && attention_fuses_q(&ca)) {
            ca.q = nullptr;
            ca.qp_a = w.xn; ca.qp_lda = D; ca.qp_k = D; ca.qp_w = folded ? bw.ca_q_w_prenorm : bw.ca_q_w; ca.qp_w_tiled = m->gemm_weights_tiled;
            if (folded) { ca.qp_row_ss = w.rowss; ca.qp_row_ss_tiles = D / 64; ca.qp_row_ss_dim = D; ca.qp_row_ss_eps = 1e-5f; }
            ca.q_norm_weight = bw.ca_q_norm_w;
        } else {
            GaGemmArgs gq{};
            gq.M = Mca; gq.N = D; gq.K = D; gq.epilogue = GA_GEMM_EPI_STORE_BF16; gq.A = w.xn; gq.lda = D;
            gq.W = folded ? bw.ca_q_w_prenorm : bw.ca_q_w; gq.w_tiled = m->gemm_weights_tiled;
            gq.out = w.qkv; gq.ldo = D;
            if (folded) { gq.row_ss = w.rowss; gq.row_ss_tiles = D / 64; gq.row_ss_dim = D; gq.row_ss_eps = 1e-5f; }
            gq.qk_w0 = bw.ca_q_norm_w; gq.qk_cols0 = D; gq.qk_cols1 = D;           // q_norm fused into the projection
            GA_UNLESS(32, ga_gemm_bf16(&gq, stream));
        }
        if (i == 0 && !join()) return GA_DIT_ERR_LAUNCH;     // the conditioning chain's results from here on (mod, tvec, block 0's shift rows)
        PrefetchJob pf{};
        {
            const unsigned DD = (unsigned)D * (unsigned)D;
            pf.ptr[0] = reinterpret_cast<const char *>(bw.fc2_w); pf.bytes[0] = 8u * DD;
            if (pf_mode >= 3) { pf.ptr[1] = reinterpret_cast<const char *>(bw.proj_w); pf.bytes[1] = 2u * DD; }
            if (pf_mode >= 4) {   // what the NEXT block's cross-attention and its output projection start on (after the last block: block 0 of
                                  // the next evaluation -- a sampling loop comes straight back)
                const int ni = i + 1 < m->depth ? i + 1 : 0;
                const GaDitBlockWeights &nb = m->blocks[ni];
                pf.ptr[2] = reinterpret_cast<const char *>(nb.ca_out_w); pf.bytes[2] = 2u * DD;
                const uint16_t *qw = (ni > 0 && can_fold(m, ni) && nb.ca_q_w_prenorm) ? nb.ca_q_w_prenorm : nb.ca_q_w;
                pf.ptr[3] = reinterpret_cast<const char *>(qw); pf.bytes[3] = 2u * DD;
                if (pf_mode == 5 && ni > 0) {         // ... and its cached K / V^T of the image tokens (the items that take part)
                    const size_t Mp_ = ((size_t)a->ctx_tokens + 63) / 64 * 64;
                    pf.ptr[4] = reinterpret_cast<const char *>(a->ca_k + (size_t)(i + 1) * B * a->ctx_tokens * D);
                    pf.bytes[4] = (unsigned)((size_t)ca_batch * a->ctx_tokens * D * 2 / 1024 * 1024);
                    pf.ptr[5] = reinterpret_cast<const char *>(a->ca_vt + (size_t)(i + 1) * B * D * Mp_);
                    pf.bytes[5] = (unsigned)((size_t)ca_batch * D * Mp_ * 2 / 1024 * 1024);
                }
            }
            // (the next evaluation's conditioning-chain weights -- adaLN 12 MB, timestep MLP -- behind the last block's grid: measured, no gain)
            if (pf_mode >= 6) {   // this block's qkv and fc1 weights once more (the shift rows' pass over them was a block ago)
                pf.ptr[4] = reinterpret_cast<const char *>(bw.qkv_w); pf.bytes[4] = 6u * DD;
                pf.ptr[5] = reinterpret_cast<const char *>(bw.fc1_w); pf.bytes[5] = 8u * DD;
            }
        }
        // round 6: EVERY block's shift rows ride behind its own cross-attention grid (as block 0's since round 5), none behind the self-attention
        // grid of the block before: the CA grid idles the same 64 CUs for 21 us instead of 13, and the weights the rows' pass pulls in are
        // used two and five launches later.  Same-box A/B (profiles/r6_sb_on_ca_ab.txt): batch 1 2.332 -> 2.272 ms per evaluation, CFG batch 2
        // 2.879 -> 2.864; GA_DIT_SB_ON_CA=0: the round-4 placement
        static const bool sb_on_ca = [] { const char *e = getenv("GA_DIT_SB_ON_CA"); return !e || atoi(e) != 0; }();
        if ((i == 0 || sb_on_ca) && sb_tail0) {
            ShiftBiasJob job{{bw.qkv_w, bw.fc1_w}, {bw.qkv_b, bw.fc1_b}, w.mod + (size_t)i * B * 6 * D, w.sbias + (size_t)i * B * 7 * D,
                             6 * (long long)D, 3 * (long long)D, 3 * D, 4 * D, D, B, m->gemm_weights_tiled};
            GA_UNLESS(2, attention_with_tail(&ca, &job, stream, pf_ca ? &pf : nullptr, pf_ca));
        } else if (pf_ca)
            GA_UNLESS(2, attention_with_tail(&ca, nullptr, stream, &pf, pf_ca));
        else
            GA_UNLESS(2, ga_attention_bf16(&ca, stream));
        GaGemmArgs go{};
        go.M = Mca; go.N = D; go.K = D; go.epilogue = GA_GEMM_EPI_RESIDUAL; go.A = w.att; go.lda = D; go.W = bw.ca_out_w; go.w_tiled = m->gemm_weights_tiled;
        go.bias = bw.ca_out_b; go.out = w.xres; go.ldo = D; go.gate = nullptr; go.rows_per_batch = L;
        const float *sbias = w.sbias + (size_t)i * B * 7 * D;   // [B][3D] qkv | [B][4D] fc1
        if (fold_mod) {
            // norm1 folded: this GEMM runs over ALL rows -- the rows of the items that skipped the cross-attention take its output bias
            // with a zero product (k_rows) -- and leaves bf16(x norm1.weight (1 + scale_msa)) in xn, the rows' sums of squares in rowss
            go.M = Mrows; go.k_rows = Mca;
            go.emit_x = w.xn; go.emit_ld = D; go.emit_ss = w.rowss; go.emit_w = bw.norm1_w; go.emit_scale = mod + 1 * D;
            go.emit_scale_stride = 6 * (int64_t)D;
        }
        GA_UNLESS(32, ga_gemm_bf16(&go, stream));
        // self-attention
        // (rows of the items that skipped the cross-attention pick up its output bias here)
        GaRmsNormArgs n1{Mrows, D, L, w.xres, bw.norm1_w, mod + 1 * D, mod + 0 * D, 6 * (int64_t)D, w.xn,
                         Mca < Mrows ? bw.ca_out_b : nullptr, Mca};
        if (!fold_mod) GA_UNLESS(4, ga_rmsnorm_modulate(&n1, stream));
        GaGemmArgs gqkv{};
        gqkv.M = Mrows; gqkv.N = 3 * D; gqkv.K = D; gqkv.epilogue = GA_GEMM_EPI_STORE_BF16; gqkv.A = w.xn; gqkv.lda = D;
        gqkv.W = bw.qkv_w; gqkv.w_tiled = m->gemm_weights_tiled; gqkv.bias = bw.qkv_b; gqkv.out = w.qkv; gqkv.ldo = 2 * D;   // q | k row-major ...
        gqkv.vt = w.vt; gqkv.vt_col0 = 2 * D; gqkv.vt_ld = Lp; gqkv.rows_per_batch = L;  // ... v transposed
        gqkv.qk_w0 = bw.q_norm_w; gqkv.qk_cols0 = D; gqkv.qk_w1 = bw.k_norm_w; gqkv.qk_cols1 = 2 * D;  // per-head q/k RMSNorm
        if (fold_mod) {
            gqkv.row_ss = w.rowss; gqkv.row_ss_tiles = D / 64; gqkv.row_ss_dim = D; gqkv.row_ss_eps = 1e-5f;
            gqkv.bias = sbias; gqkv.bias_stride = 3 * (int64_t)D;
        }
        gqkv.splitk_ws = w.splitk; gqkv.splitk_ws_bytes = (int64_t)w.splitk_bytes;
        GA_UNLESS(16, ga_gemm_bf16(&gqkv, stream));
        GaAttentionArgs sa{B, m->heads, L, L, w.qkv, w.qkv + D, w.vt, 2 * D, 2 * D, Lp, nullptr, nullptr, w.att, D};
        static const bool sb_on_ca2 = [] { const char *e = getenv("GA_DIT_SB_ON_CA"); return !e || atoi(e) != 0; }();
        if (fold_mod && sb_tail && i + 1 < m->depth && !(sb_on_ca2 && sb_tail0)) {
            const GaDitBlockWeights &nb = m->blocks[i + 1];
            ShiftBiasJob job{{nb.qkv_w, nb.fc1_w}, {nb.qkv_b, nb.fc1_b}, w.mod + (size_t)(i + 1) * B * 6 * D, w.sbias + (size_t)(i + 1) * B * 7 * D,
                             6 * (long long)D, 3 * (long long)D, 3 * D, 4 * D, D, B, m->gemm_weights_tiled};
            GA_UNLESS(1, attention_with_tail(&sa, &job, stream, pf_sa ? &pf : nullptr, pf_sa));
        } else if (pf_sa)
            GA_UNLESS(1, attention_with_tail(&sa, nullptr, stream, &pf, pf_sa));
        else
            GA_UNLESS(1, ga_attention_bf16(&sa, stream));
        GaGemmArgs gp{};
        gp.M = Mrows; gp.N = D; gp.K = D; gp.epilogue = GA_GEMM_EPI_RESIDUAL; gp.A = w.att; gp.lda = D; gp.W = bw.proj_w; gp.w_tiled = m->gemm_weights_tiled;
        gp.bias = bw.proj_b; gp.out = w.xres; gp.ldo = D; gp.gate = mod + 2 * D; gp.gate_stride = 6 * (int64_t)D;
        gp.rows_per_batch = L;
        if (fold_mod) {   // norm2 folded the same way: proj emits, fc1 consumes
            gp.emit_x = w.xn; gp.emit_ld = D; gp.emit_ss = w.rowss; gp.emit_w = bw.norm2_w; gp.emit_scale = mod + 4 * D;
            gp.emit_scale_stride = 6 * (int64_t)D;
        }
        GA_UNLESS(16, ga_gemm_bf16(&gp, stream));
        // FusedMLP
        GaRmsNormArgs n2{Mrows, D, L, w.xres, bw.norm2_w, mod + 4 * D, mod + 3 * D, 6 * (int64_t)D, w.xn, nullptr, 0};
        if (!fold_mod) GA_UNLESS(4, ga_rmsnorm_modulate(&n2, stream));
        GaGemmArgs g1{};
        g1.M = Mrows; g1.N = 4 * D; g1.K = D; g1.epilogue = GA_GEMM_EPI_GELU_BF16; g1.A = w.xn; g1.lda = D; g1.W = bw.fc1_w; g1.w_tiled = m->gemm_weights_tiled;
        g1.bias = bw.fc1_b; g1.out = w.hmid; g1.ldo = 4 * D;
        if (fold_mod) {
            g1.row_ss = w.rowss; g1.row_ss_tiles = D / 64; g1.row_ss_dim = D; g1.row_ss_eps = 1e-5f;
            g1.bias = sbias + (size_t)B * 3 * D; g1.bias_stride = 4 * (int64_t)D; g1.rows_per_batch = L;
        }
        g1.splitk_ws = w.splitk; g1.splitk_ws_bytes = (int64_t)w.splitk_bytes;
        GA_UNLESS(8, ga_gemm_bf16(&g1, stream));
        GaGemmArgs g2{};
        g2.M = Mrows; g2.N = D; g2.K = 4 * D; g2.epilogue = GA_GEMM_EPI_RESIDUAL; g2.A = w.hmid; g2.lda = 4 * D;
        g2.W = bw.fc2_w; g2.w_tiled = m->gemm_weights_tiled; g2.bias = bw.fc2_b; g2.out = w.xres; g2.ldo = D; g2.gate = mod + 5 * D;
        g2.gate_stride = 6 * (int64_t)D; g2.rows_per_batch = L;
        if (i + 1 < m->depth && can_fold(m, i + 1)) { g2.emit_x = w.xn; g2.emit_ld = D; g2.emit_ss = w.rowss; }
        g2.splitk_ws = w.splitk; g2.splitk_ws_bytes = (int64_t)w.splitk_bytes;
        GA_UNLESS(8, ga_gemm_bf16(&g2, stream));
    }
    {
        FinalArgs f{Mrows, D, m->out_channels, L, w.xres, m->final_table, w.tvec, m->final_w, m->final_b, a->out,
                    0.f, 0, nullptr, nullptr, nullptr, 0, nullptr};
        int rows = Mrows;
        if (const GaDitSamplerStep *st = a->step) {
            if (st->velocity) {     // the (guided) velocity alone: dt == nullptr tells the kernel
                if (st->cfg && (B % 2 != 0)) return GA_DIT_ERR_BAD_SHAPE;
                f.cfg_scale = st->cfg_scale; f.cfg = st->cfg ? 1 : 0; f.state = st->velocity;
            } else {
                if (!st->dt || !st->state || !st->counter || (st->cfg && (B % 2 != 0)) || st->state != a->x ||
                    m->out_channels != m->in_channels)
                    return GA_DIT_ERR_BAD_SHAPE;
                f.cfg_scale = st->cfg_scale; f.cfg = st->cfg ? 1 : 0; f.dt = st->dt; f.state = st->state; f.traj = st->traj;
                f.traj_stride = st->traj_stride; f.counter = st->counter;
            }
            rows = st->cfg ? Mrows / 2 : Mrows;
        }
        // (with CFG a workgroup serves two rows of the conditional half: two waves each for the rows and their unconditional twins)
        const size_t wbytes = (size_t)m->out_channels * D * sizeof(float);
        const int w_in_lds = wbytes <= 60 * 1024 ? 1 : 0;      // (the default dynamic LDS limit; beyond it the rows read the weight from the L2 as before)
        hipLaunchKernelGGL(final_layer_kernel, dim3(a->step && a->step->cfg ? (rows + 1) / 2 : (rows + 3) / 4), dim3(256), w_in_lds ? wbytes : 0, s, f, w_in_lds);
    }
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}

extern "C" int ga_dit_sampler_advance(int32_t *counter, const float *t_grid, const float *dt_grid, int32_t grid_len,
                                      float *timesteps, int32_t batch, float *dt, void *stream)
{
    if (!counter || !t_grid || !dt_grid || !timesteps || !dt) return GA_DIT_ERR_NULL_ARG;
    if (grid_len <= 0 || batch <= 0 || batch > 64) return GA_DIT_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(gadit::sampler_advance_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), counter, t_grid,
                       dt_grid, grid_len, timesteps, batch, dt);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}
