// dit_attention_hd.hip -- fused multi-head attention forward for head dims OTHER than 64 (no mask, no dropout), bf16 MFMA, gfx950.
//
// The released denoisers have heads of 64 (dit_attention.hip: LDS-DMA rings, 128-byte tile rows).  The reference registry also holds
// DiT-PixArt-PCD-CLAY-XL (/root/reference/dit/dit_i23d.py:1526-1535, 1677: width 1152, 16 heads -> head dim 72), unreleased; this file
// is what makes that entry run: the same product formulation as the 64-wide kernel -- S^T = K Q^T with the K fragment rows permuted so
// that a lane's P values of a 32-key block are 8 consecutive keys, O^T = V^T P^T, softmax state lane-local -- with the head dim padded
// to the MFMA shapes (QK^T: k-steps of 32 -> 96 columns, zero beyond 72; PV: d tiles of 16 -> 5 tiles, rows beyond 72 not stored),
// tiles staged through registers into padded LDS rows (V transposed on the way: the projection GEMM's V^T store is 64-wide too), one
// key tile per barrier pair.  A correctness-first path (MemEffAttention / MemoryEfficientCrossAttention between their projections,
// /root/reference/vit/vision_transformer.py:284-297, ldm/modules/attention.py:514-548); q and k arrive already RMS-normalised
// (ga_head_rmsnorm_bf16 below: the GEMM epilogue's per-head norm is 64-wide as well).
#include <stdlib.h>

#include "dit_common.h"

namespace gadit {

__device__ __forceinline__ float group_max_hd(float t)   // over lanes l, l^16, l^32, l^48 (the four lane groups of one query)
{
    const unsigned u = __float_as_uint(t);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned v = __float_as_uint(m);
    const auto c = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}

// HDP: head dim rounded up to a multiple of 32.  Workgroup = 4 waves x 16 queries; grid (ceil(Lq / 64), heads, batch).
template <int HDP>
__global__ __launch_bounds__(256) void attention_hd_kernel(GaAttentionHdArgs a)
{
    constexpr int KB = 64, KROW = HDP + 8, VROW = KB + 8, NK = HDP / 32, ND = HDP / 16;
    __shared__ __attribute__((aligned(16))) uint16_t Ks[KB * KROW];    // [key][d], d >= head_dim zero
    __shared__ __attribute__((aligned(16))) uint16_t Vt[HDP * VROW];   // [d][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int hd = a.head_dim, cpr = hd >> 3;                          // 16-byte chunks per row
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64 + wave * 16;
    const int Lq = a.Lq, Lk = a.Lk;
    for (int i = tid; i < KB * KROW; i += 256) Ks[i] = 0;
    for (int i = tid; i < HDP * VROW; i += 256) Vt[i] = 0;

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + c16][kk*32 + g*8 .. +7] * head_dim^-1/2 * log2(e), zero beyond head_dim
    bf16x8 qf[NK];
    {
        const int row = min(q0 + c16, Lq - 1);
        const uint16_t *qp = a.q + ((size_t)b * Lq + row) * a.q_stride + (size_t)h * hd;
        const float rs = rsqrtf((float)hd) * 1.4426950408889634f;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (kk * 4 + g < cpr) raw = *reinterpret_cast<const uint4 *>(qp + kk * 32 + g * 8);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qf[kk][2 * e] = (short)f32_to_bf16(__uint_as_float(w[e] << 16) * rs);
                qf[kk][2 * e + 1] = (short)f32_to_bf16(__uint_as_float(w[e] & 0xffff0000u) * rs);
            }
        }
    }
    f32x4 o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;
    const uint16_t *kb_ = a.k + (size_t)b * Lk * a.k_stride + (size_t)h * hd;
    const uint16_t *vb_ = a.v + (size_t)b * Lk * a.v_stride + (size_t)h * hd;
    const int ntiles = (Lk + KB - 1) / KB, nchunk = KB * cpr;
    const int krow = 8 * (c16 >> 2) + (c16 & 3);
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                       // everybody has finished the previous tile (and the zero fill)
        for (int c = tid; c < nchunk; c += 256) {
            const int row = c / cpr, part = c - row * cpr;
            const int key = min(t * KB + row, Lk - 1);     // keys beyond the end are masked below
            const uint4 kv = *reinterpret_cast<const uint4 *>(kb_ + (size_t)key * a.k_stride + part * 8);
            const uint4 vv = *reinterpret_cast<const uint4 *>(vb_ + (size_t)key * a.v_stride + part * 8);
            *reinterpret_cast<uint4 *>(Ks + row * KROW + part * 8) = kv;
            const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(part * 8 + 2 * e) * VROW + row] = (uint16_t)(w[e] & 0xffffu);
                Vt[(part * 8 + 2 * e + 1) * VROW + row] = (uint16_t)(w[e] >> 16);
            }
        }
        __syncthreads();
        // S^T = K Q^T : s[kf][r] <-> key 32 (kf >> 1) + 8 g + 4 (kf & 1) + r of the tile, query c16
        f32x4 s[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) s[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const bf16x8 kfrag = *reinterpret_cast<const bf16x8 *>(Ks + ((kf >> 1) * 32 + (kf & 1) * 4 + krow) * KROW + kk * 32 + g * 8);
                s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[kk], s[kf], 0, 0, 0);
            }
        const int kbase = t * KB + g * 8;
        float tmax = -1e30f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kbase + (kf >> 1) * 32 + (kf & 1) * 4 + r >= Lk) s[kf][r] = -1e30f;
                tmax = fmaxf(tmax, s[kf][r]);
            }
        tmax = group_max_hd(tmax);
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        bf16x8 pf[2];
        float psum = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[kf][r] - m_new);
                psum += p;
                pf[kf >> 1][(kf & 1) * 4 + r] = (short)f32_to_bf16(p);
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int df = 0; df < ND; ++df) { o[df][0] *= alpha; o[df][1] *= alpha; o[df][2] *= alpha; o[df][3] *= alpha; }
        // O^T += V^T P^T : o[df][r] = O[q = c16][d = df*16 + g*4 + r]; the lane's 8 P of block kb are keys 32 kb + 8 g ..
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int df = 0; df < ND; ++df) {
                const bf16x8 vfrag = *reinterpret_cast<const bf16x8 *>(Vt + (df * 16 + c16) * VROW + kb * 32 + g * 8);
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pf[kb], o[df], 0, 0, 0);
            }
    }
    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int row = q0 + c16;
    if (row < Lq) {
        uint16_t *op = a.out + ((size_t)b * Lq + row) * a.out_stride + (size_t)h * hd + g * 4;
#pragma unroll
        for (int df = 0; df < ND; ++df)
            if (df * 16 + g * 4 < hd)
                *reinterpret_cast<uint2 *>(op + df * 16) = make_uint2(pack_bf16x2(o[df][0] * inv, o[df][1] * inv), pack_bf16x2(o[df][2] * inv, o[df][3] * inv));
    }
}

// x[r][h][:] <- x[r][h][:] * rsqrt(mean(x[r][h][:]^2) + 1e-5) * w[:], one wavefront per (row, head); head_dim <= 128
__global__ __launch_bounds__(256) void head_rmsnorm_kernel(uint16_t *__restrict__ x, int64_t rows, int64_t row_stride, int heads, int hd,
                                                            const float *__restrict__ w)
{
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= rows * heads) return;
    const int64_t r = item / heads;
    const int h = (int)(item - r * heads);
    uint16_t *p = x + r * row_stride + (int64_t)h * hd;
    const float v0 = lane < hd ? bf16_to_f32(p[lane]) : 0.f, v1 = lane + 64 < hd ? bf16_to_f32(p[lane + 64]) : 0.f;
    const float ss = wave_sum(v0 * v0 + v1 * v1);
    const float rs = rsqrtf(ss / (float)hd + 1e-5f);
    if (lane < hd) p[lane] = f32_to_bf16(v0 * (rs * w[lane]));
    if (lane + 64 < hd) p[lane + 64] = f32_to_bf16(v1 * (rs * w[lane + 64]));
}

}  // namespace gadit

extern "C" int ga_attention_hd_bf16(const GaAttentionHdArgs *a, void *stream)
{
    using namespace gadit;
    if (!a || !a->q || !a->k || !a->v || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->batch <= 0 || a->heads <= 0 || a->Lq <= 0 || a->Lk <= 0 || a->head_dim < 8 || a->head_dim > 128 || a->head_dim % 8 ||
        a->q_stride % 8 || a->k_stride % 8 || a->v_stride % 8 || a->out_stride % 4)
        return GA_DIT_ERR_BAD_SHAPE;
    if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v) % 16 != 0 || (uintptr_t)a->out % 8 != 0) return GA_DIT_ERR_BAD_SHAPE;
    const dim3 grid((unsigned)((a->Lq + 63) / 64), (unsigned)a->heads, (unsigned)a->batch);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int hdp = (a->head_dim + 31) / 32 * 32;
    if (hdp == 32) hipLaunchKernelGGL(attention_hd_kernel<32>, grid, dim3(256), 0, s, *a);
    else if (hdp == 64) hipLaunchKernelGGL(attention_hd_kernel<64>, grid, dim3(256), 0, s, *a);
    else if (hdp == 96) hipLaunchKernelGGL(attention_hd_kernel<96>, grid, dim3(256), 0, s, *a);
    else hipLaunchKernelGGL(attention_hd_kernel<128>, grid, dim3(256), 0, s, *a);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}

extern "C" int ga_head_rmsnorm_bf16(ga_bf16 *x, int64_t rows, int64_t row_stride, int32_t heads, int32_t head_dim, const float *weight,
                                    void *stream)
{
    using namespace gadit;
    if (!x || !weight) return GA_DIT_ERR_NULL_ARG;
    if (rows <= 0 || heads <= 0 || head_dim <= 0 || head_dim > 128 || row_stride < (int64_t)heads * head_dim) return GA_DIT_ERR_BAD_SHAPE;
    const int64_t items = rows * heads;
    hipLaunchKernelGGL(head_rmsnorm_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, rows,
                       row_stride, heads, head_dim, weight);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}
