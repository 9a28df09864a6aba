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
// One wave per (group, head); lane = head dimension.  S*S scores are wave sums, the softmax and the S*S weighted sums of
// V are lane-local.  At S = 4 ... 9 this is a few hundred instructions per wave and the kernel streams qkv once.
__global__ __launch_bounds__(256) void tiny_attention_kernel(GaTinyAttentionArgs a)
{
    const long long gh = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (gh >= (long long)a.groups * a.heads) return;
    const int grp = (int)(gh / a.heads), h = (int)(gh - (long long)grp * a.heads);
    const int S = a.S, C = a.heads * 64;
    const uint16_t *base = a.qkv + (size_t)grp * S * 3 * C + h * 64 + lane;
    float q[16], k[16], v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < S) {
            const uint16_t *r = base + (size_t)i * 3 * C;
            q[i] = bf16_to_f32(r[0]);
            k[i] = bf16_to_f32(r[C]);
            v[i] = bf16_to_f32(r[2 * C]);
        }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < S) {
            float sc[16], mx = -1e30f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < S) {
                    sc[j] = wave_sum(q[i] * k[j]) * 0.125f;
                    mx = fmaxf(mx, sc[j]);
                }
            float den = 0.f, o = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < S) {
                    const float pj = __expf(sc[j] - mx);
                    den += pj;
                    o += pj * v[j];
                }
            a.out[((size_t)grp * S + i) * C + h * 64 + lane] = f32_to_bf16(o / den);
        }
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
    float mean = 0.f, rs = 1.f;
    if (a.mode == 1) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += x[d];
        mean = wave_sum(s) / (float)D;
        float q = 0.f;
        for (int d = lane; d < D; d += 64) { const float e = x[d] - mean; q += e * e; }
        rs = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
    }
    float acc[13];
#pragma unroll
    for (int c = 0; c < 13; ++c) acc[c] = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float y = a.mode == 1 ? (x[d] - mean) * rs * a.ln_weight[d] + a.ln_bias[d] : silu_f(x[d]);
#pragma unroll
        for (int c = 0; c < 13; ++c) acc[c] += y * a.w[(size_t)c * D + d];
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
    GA_LAUNCH_ROWS(tiny_attention_kernel, (long long)a->groups * a->heads, a, stream);
}

extern "C" int ga_surfel_head(const GaSurfelHeadArgs *a, void *stream)
{
    if (!a || !a->x || !a->w || !a->b || !a->anchor || !a->gaussians || !a->pre_out) return GA_DIT_ERR_NULL_ARG;
    if (a->mode == 1 && (!a->ln_weight || !a->ln_bias || !a->base_pre || a->f < 1 || a->rows % a->f)) return GA_DIT_ERR_NULL_ARG;
    if (a->rows < 0 || a->D < 1 || (a->mode != 0 && a->mode != 1)) return GA_DIT_ERR_BAD_SHAPE;
    GA_LAUNCH_ROWS(surfel_head_kernel, a->rows, a, stream);
}
