// decode_ops.hip -- row kernels of the surfel decode (include/ga_decode.h), gfx950.  Everything here is HBM-bound: one
// wave per row (or per group-head), 16-byte accesses where the layout allows; the GEMMs and the 768-token attention of
// the decoder backbone are the DiT kernels (dit_gemm.hip, dit_attention.hip).
#include "dit_common.h"

#include "../../include/ga_decode.h"

namespace gadit {

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float gelu_tanh_f(float v)
{
    return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v)));
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tiny_mlp_silu_kernel(GaTinyMlpArgs a)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.M) return;
    float h[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float s = 0.f;
        if (j < a.Ch) {
            s = a.b1[j];
            for (int k = 0; k < a.Cin; ++k) s += a.w1[j * a.Cin + k] * a.x[(size_t)row * a.Cin + k];
            s = gelu_tanh_f(s);
        }
        h[j] = s;
    }
    for (int d = lane; d < a.D; d += 64) {
        float s = a.b2[d];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < a.Ch) s += a.w2[(size_t)d * a.Ch + j] * h[j];
        a.out[(size_t)row * a.D + d] = f32_to_bf16(silu_f(s));
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_modulate_kernel(GaLayerNormArgs a)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.M) return;
    const int D = a.D;
    const float *x = a.x + (size_t)row * D;
    const float *sc = a.scale ? a.scale + (size_t)row * a.mod_stride : nullptr;
    const float *sh = a.shift ? a.shift + (size_t)row * a.mod_stride : nullptr;
    // lane owns the float4 at d = c*256 + lane*4 of every 256-wide chunk c; all operands are requested up front
    float4 v[8], w[8], bb[8], s4[8], h4[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int d = c * 256 + lane * 4;
        if (d < D) {
            v[c] = *reinterpret_cast<const float4 *>(x + d);
            if (a.weight) {
                w[c] = *reinterpret_cast<const float4 *>(a.weight + d);
                bb[c] = *reinterpret_cast<const float4 *>(a.bias + d);
            }
            if (sc) {
                s4[c] = *reinterpret_cast<const float4 *>(sc + d);
                h4[c] = *reinterpret_cast<const float4 *>(sh + d);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c * 256 + lane * 4 < D) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c * 256 + lane * 4 < D) {
            const float e0 = v[c].x - mean, e1 = v[c].y - mean, e2 = v[c].z - mean, e3 = v[c].w - mean;
            q += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
        }
    const float rs = rsqrtf(wave_sum(q) / (float)D + a.eps);
    uint16_t *o = a.out + (size_t)row * D;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int d = c * 256 + lane * 4;
        if (d < D) {
            float y[4] = {(v[c].x - mean) * rs, (v[c].y - mean) * rs, (v[c].z - mean) * rs, (v[c].w - mean) * rs};
            if (a.weight) {
                y[0] = y[0] * w[c].x + bb[c].x; y[1] = y[1] * w[c].y + bb[c].y;
                y[2] = y[2] * w[c].z + bb[c].z; y[3] = y[3] * w[c].w + bb[c].w;
            }
            if (sc) {
                y[0] = y[0] * (1.f + s4[c].x) + h4[c].x; y[1] = y[1] * (1.f + s4[c].y) + h4[c].y;
                y[2] = y[2] * (1.f + s4[c].z) + h4[c].z; y[3] = y[3] * (1.f + s4[c].w) + h4[c].w;
            }
            *reinterpret_cast<uint2 *>(o + d) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_tokens_kernel(GaAssembleArgs a)
{
    const int S = 1 + a.f;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= (long long)a.P * S) return;
    const int p = (int)(row / S), j = (int)(row - (long long)p * S);
    const float *src;
    if (j == 0) {
        const long long sr = a.src_f ? (long long)(p / a.src_f) * (1 + a.src_f) + 1 + p % a.src_f : p;
        src = a.src + sr * a.D;
    } else {
        src = a.latent_embedding + (size_t)(j - 1) * a.D;
    }
    float *dst = a.out + row * a.D;
    for (int d = lane * 4; d < a.D; d += 256) *reinterpret_cast<float4 *>(dst + d) = *reinterpret_cast<const float4 *>(src + d);
}

// ---------------------------------------------------------------------------------------------------------------
// Tiny-sequence attention on the matrix cores.  One wave per (head, tile of G = 16 / S whole groups): the tile's <= 16
// rows are both the queries and the keys of ONE 16x16 score block, attention between different groups is masked out
// (block-diagonal), so 4 groups of 4 / 3 groups of 5 / 1 group of 9 rows cost two v_mfma_f32_16x16x32_bf16 for the
// scores and four for P V.  The layout follows dit_attention.hip: the swapped product S^T = K Q^T leaves query c in
// lanes {c, c+16, c+32, c+48} with four keys each, so the softmax needs two xor-shuffles, and those four probabilities
// ARE the lane's B-operand of O^T = V^T P^T (keys 4g..4g+3 of the k-block, the other 28 k-slots are zero) -- P never
// moves.  V^T fragments are gathered straight from global memory (4 bf16 per lane and 16-wide d block; the lines are
// shared by the whole wave).  The first version (one wave per (group, head), lane = head dimension, S^2 wave sums)
// took 272 us per call at the release sizes.
__global__ __launch_bounds__(256) void tiny_attention_kernel(GaTinyAttentionArgs a)
{
    const int S = a.S, G = 16 / S, C = a.heads * 64;
    const long long ntile = (a.groups + G - 1) / G;
    const long long wt = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wt >= ntile * a.heads) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const long long tile = wt / a.heads;
    const int h = (int)(wt - tile * a.heads);
    const long long row0 = tile * G * S;
    const int nrow = (int)min((long long)G, a.groups - tile * G) * S;   // valid rows of this tile
    const uint16_t *base = a.qkv + (size_t)row0 * 3 * C + h * 64;
    const int rc = min(c, nrow - 1);                                      // padding rows re-read a valid one
    // S^T = K Q^T: A = K rows (key = lane & 15), B = Q rows (query = lane & 15), k-chunk (lane >> 4) * 8 of each 32
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8 *>(base + (size_t)rc * 3 * C + kk * 32 + g * 8);
        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(base + (size_t)rc * 3 * C + C + kk * 32 + g * 8);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, s, 0, 0, 0);
    }
    // lane (g, c): s[r] = score(query c, key 4g + r); keys of another group or past the tile are masked
    const int qgrp = c / S;
    float p[4], mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = 4 * g + r;
        const bool ok = key < nrow && key / S == qgrp;
        p[r] = ok ? s[r] * 0.125f : -1e30f;
        mx = fmaxf(mx, p[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float den = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        p[r] = p[r] > -1e29f ? __expf(p[r] - mx) : 0.f;
        den += p[r];
    }
    den += __shfl_xor(den, 16, 64);
    den += __shfl_xor(den, 32, 64);
    const float inv = 1.0f / den;
    bf16x8 pf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pf[r] = (short)f32_to_bf16(p[r] * inv);
        pf[4 + r] = 0;
    }
    // O^T = V^T P^T per 16-wide d block: A = V^T rows (d = db*16 + (lane & 15)), k-slot e <-> key 16*(e>>2) + 4g + (e&3)
    const uint16_t *vb = base + 2 * C;
    uint16_t *out = a.out + (size_t)(row0 + c) * C + h * 64 + 4 * g;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        bf16x8 vf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = min(4 * g + r, nrow - 1);        // its probability is zero when it is padding
            vf[r] = (short)vb[(size_t)key * 3 * C + db * 16 + c];
            vf[4 + r] = 0;
        }
        f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        // lane (g, c): o[r] = O[query c][d = db*16 + 4g + r]
        if (c < nrow)
            *reinterpret_cast<uint2 *>(out + db * 16) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void surfel_head_kernel(GaSurfelHeadArgs a)
{
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= a.rows) return;
    const int D = a.D;
    long long xr = row, p = row;
    if (a.mode == 1) {
        p = row / a.f;
        xr = p * (1 + a.f) + 1 + (row - p * a.f);
    }
    const float *x = a.x + xr * D;
    // the row lives in registers (float4 at d = c*256 + lane*4, D <= 2048): one pass over HBM
    float4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c * 256 + lane * 4 < D) v[c] = *reinterpret_cast<const float4 *>(x + c * 256 + lane * 4);
    float mean = 0.f, rs = 1.f;
    if (a.mode == 1) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c * 256 + lane * 4 < D) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
        mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (c * 256 + lane * 4 < D) {
                const float e0 = v[c].x - mean, e1 = v[c].y - mean, e2 = v[c].z - mean, e3 = v[c].w - mean;
                q += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
            }
        rs = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
    }
    float acc[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) acc[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int d = c * 256 + lane * 4;
        if (d < D) {
            float y[4] = {v[c].x, v[c].y, v[c].z, v[c].w};
            if (a.mode == 1) {
                const float4 lw = *reinterpret_cast<const float4 *>(a.ln_weight + d), lb = *reinterpret_cast<const float4 *>(a.ln_bias + d);
                y[0] = (y[0] - mean) * rs * lw.x + lb.x; y[1] = (y[1] - mean) * rs * lw.y + lb.y;
                y[2] = (y[2] - mean) * rs * lw.z + lb.z; y[3] = (y[3] - mean) * rs * lw.w + lb.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = silu_f(y[e]);
            }
#pragma unroll
            for (int o = 0; o < 13; ++o) {
                const float4 w4 = *reinterpret_cast<const float4 *>(a.w + (size_t)o * D + d);
                acc[o] += (y[0] * w4.x + y[1] * w4.y) + (y[2] * w4.z + y[3] * w4.w);
            }
        }
    }
    float pre[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) pre[c] = wave_sum(acc[c]) + a.b[c];
    if (lane != 0) return;
    float pos[3];
    if (a.mode == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) pos[c] = tanhf(pre[c]) * 0.225f * a.skip_weight + a.anchor[row * 3 + c];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) pos[c] = tanhf(pre[c]) * 0.225f + a.anchor[p * 13 + c];
#pragma unroll
        for (int c = 0; c < 13; ++c) pre[c] += a.base_pre[p * 13 + c];
    }
    float *g = a.gaussians + row * 13, *po = a.pre_out + row * 13;
#pragma unroll
    for (int c = 0; c < 13; ++c) po[c] = pre[c];
    g[0] = pos[0]; g[1] = pos[1]; g[2] = pos[2];
    g[3] = 1.0f / (1.0f + expf(-pre[3]));
    const float sf = 0.0045f / 0.6931471805599453f;  // scene_extent / softplus(0)
#pragma unroll
    for (int c = 4; c < 6; ++c) g[c] = (pre[c] > 20.f ? pre[c] : log1pf(expf(pre[c]))) * sf;  // F.softplus (threshold 20)
    const float n2 = pre[6] * pre[6] + pre[7] * pre[7] + pre[8] * pre[8] + pre[9] * pre[9];
    const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);                                        // F.normalize eps
#pragma unroll
    for (int c = 6; c < 10; ++c) g[c] = pre[c] * inv;
#pragma unroll
    for (int c = 10; c < 13; ++c) g[c] = 0.5f * tanhf(pre[c]) + 0.5f;
}

}  // namespace gadit

using namespace gadit;

#define GA_LAUNCH_ROWS(kernel, rows, args, stream)                                                            \
    do {                                                                                                      \
        const long long r_ = (rows);                                                                          \
        if (r_ > 0)                                                                                           \
            hipLaunchKernelGGL(kernel, dim3((unsigned)((r_ + 3) / 4)), dim3(256), 0,                          \
                               reinterpret_cast<hipStream_t>(stream), *(args));                               \
        return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;                               \
    } while (0)

extern "C" int ga_tiny_mlp_silu(const GaTinyMlpArgs *a, void *stream)
{
    if (!a || !a->x || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->M < 0 || a->Cin < 1 || a->Cin > 16 || a->Ch < 1 || a->Ch > 16 || a->D < 1) return GA_DIT_ERR_BAD_SHAPE;
    GA_LAUNCH_ROWS(tiny_mlp_silu_kernel, a->M, a, stream);
}

extern "C" int ga_layernorm_modulate(const GaLayerNormArgs *a, void *stream)
{
    if (!a || !a->x || !a->out) return GA_DIT_ERR_NULL_ARG;
    if ((a->weight == nullptr) != (a->bias == nullptr) || (a->scale == nullptr) != (a->shift == nullptr)) return GA_DIT_ERR_NULL_ARG;
    if (a->M < 0 || a->D < 4 || a->D % 4 || a->D > 2048 || (a->scale && a->mod_stride % 4)) return GA_DIT_ERR_BAD_SHAPE;
    GA_LAUNCH_ROWS(layernorm_modulate_kernel, a->M, a, stream);
}

extern "C" int ga_assemble_tokens(const GaAssembleArgs *a, void *stream)
{
    if (!a || !a->src || !a->latent_embedding || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->P < 0 || a->f < 1 || a->D < 4 || a->D % 4 || a->src_f < 0 || (a->src_f && a->P % a->src_f)) return GA_DIT_ERR_BAD_SHAPE;
    GA_LAUNCH_ROWS(assemble_tokens_kernel, (long long)a->P * (1 + a->f), a, stream);
}

extern "C" int ga_tiny_attention(const GaTinyAttentionArgs *a, void *stream)
{
    if (!a || !a->qkv || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->groups < 0 || a->S < 1 || a->S > 16 || a->heads < 1) return GA_DIT_ERR_BAD_SHAPE;
    GA_LAUNCH_ROWS(tiny_attention_kernel, (long long)((a->groups + 16 / a->S - 1) / (16 / a->S)) * a->heads, a, stream);
}

extern "C" int ga_surfel_head(const GaSurfelHeadArgs *a, void *stream)
{
    if (!a || !a->x || !a->w || !a->b || !a->anchor || !a->gaussians || !a->pre_out) return GA_DIT_ERR_NULL_ARG;
    if (a->mode == 1 && (!a->ln_weight || !a->ln_bias || !a->base_pre || a->f < 1 || a->rows % a->f)) return GA_DIT_ERR_NULL_ARG;
    if (a->rows < 0 || a->D < 4 || a->D % 4 || a->D > 2048 || (a->mode != 0 && a->mode != 1)) return GA_DIT_ERR_BAD_SHAPE;
    GA_LAUNCH_ROWS(surfel_head_kernel, a->rows, a, stream);
}
