// dit_attention.hip -- fused multi-head attention forward (no mask, no dropout), head_dim 64, bf16 MFMA, gfx950.
//
// Computes what the reference obtains from xformers.ops.memory_efficient_attention between the projections of
//   MemEffAttention.forward               /root/reference/vit/vision_transformer.py:284-297  (self-attention, L = 768)
//   MemoryEfficientCrossAttention.forward /root/reference/ldm/modules/attention.py:514-548   (image tokens, Lk = 1369)
// INCLUDING the per-head RMSNorm of q and k that precedes it (learned weight[64], eps 1e-5, dit/norm.py:29-43), which
// is applied while the tiles are staged, and the softmax scale 64^-1/2.
//
// MI355X mapping (flash-style, one pass over the keys, online softmax in fp32):
//   * grid (ceil(Lq/64), heads, batch); 4 waves per workgroup, each wave owns 16 query rows, the workgroup shares
//     the staged K / V tiles (64 keys) in LDS;
//   * "swapped" products so that every reduction is lane-local or a 2-step lane shuffle: S^T = K Q^T puts one query
//     in a lane (column lane&15) with 4 keys per accumulator fragment, and O^T = V^T P^T keeps that query in the same
//     lane, so the running max / sum / rescale never cross lanes except for two xor-shuffles per tile;
//   * P never leaves registers: the PV product defines its own key order inside each 32-key block (lane-group g,
//     element e  <->  key 16*(e>>2) + 4*g + (e&3)), which is exactly how the S^T accumulators already sit in the lane;
//     V is staged TRANSPOSED in that same order, so its fragments are plain 16-byte LDS reads;
//   * K rows are RMS-normalised by the 8 lanes that stage a row (3 xor-shuffles), Q rows by the 4 lane-groups that
//     hold a row's fragments; the softmax scale and log2(e) are folded into Q so the exponentials are bare v_exp_f32.
#include "dit_common.h"

namespace gadit {

constexpr int QB = 64, KB = 64, HD = 64, LDK = HD + 8, LDV = KB + 8;

__global__ __launch_bounds__(256) void attention_fwd_kernel(GaAttentionArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t sK[KB * LDK];   // [key][d]
    __shared__ __attribute__((aligned(16))) uint16_t sVt[HD * LDV];  // [d][permuted key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wave * 16;
    const int Lq = a.Lq, Lk = a.Lk;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + c16][kk*32 + g*8 .. +7], normalised and scaled
    bf16x8 qf[2];
    {
        const int qi = min(q0 + c16, Lq - 1);
        const uint16_t *qp = a.q + ((size_t)b * Lq + qi) * a.q_stride + h * HD;
        float qv[16];
        float ss = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(qp + kk * 32 + g * 8);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qv[kk * 8 + 2 * e] = __uint_as_float(w[e] << 16);
                qv[kk * 8 + 2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) ss += qv[e] * qv[e];
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        float rs = 0.125f * 1.4426950408889634f;  // 64^-1/2 * log2(e)
        if (a.q_norm_weight) rs *= rsqrtf(ss * (1.0f / HD) + 1e-5f);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = qv[kk * 8 + e];
                if (a.q_norm_weight) {
                    // the reference rounds the normalised q to bf16 (RMSNorm returns the input dtype) before the
                    // attention kernel scales it; scaling first only moves one rounding
                    v *= a.q_norm_weight[kk * 32 + g * 8 + e];
                }
                qf[kk][e] = (short)f32_to_bf16(v * rs);
            }
        }
    }

    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;

    // staging assignment: K and V tiles are 512 16-byte chunks each, 2 per thread; chunk c -> key row c>>3, d-part c&7
    const int ntiles = (Lk + KB - 1) / KB;
    uint4 rk[2], rv[2];
    auto issue = [&](int tile) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i;
            const int key = min(tile * KB + (c >> 3), Lk - 1);
            const size_t row = (size_t)b * Lk + key;
            rk[i] = *reinterpret_cast<const uint4 *>(a.k + row * a.k_stride + h * HD + (c & 7) * 8);
            rv[i] = *reinterpret_cast<const uint4 *>(a.v + row * a.v_stride + h * HD + (c & 7) * 8);
        }
    };
    issue(0);

    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();  // everyone is done reading the previous tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + 256 * i, key = c >> 3, part = c & 7;
            // K: RMS-normalise the row (8 consecutive lanes hold it), store [key][d]
            const uint32_t kw[4] = {rk[i].x, rk[i].y, rk[i].z, rk[i].w};
            float kv[8];
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kv[2 * e] = __uint_as_float(kw[e] << 16);
                kv[2 * e + 1] = __uint_as_float(kw[e] & 0xffff0000u);
            }
            if (a.k_norm_weight) {
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += kv[e] * kv[e];
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                ss += __shfl_xor(ss, 4, 64);
                const float rs = rsqrtf(ss * (1.0f / HD) + 1e-5f);
#pragma unroll
                for (int e = 0; e < 8; ++e) kv[e] *= rs * a.k_norm_weight[part * 8 + e];
            }
            uint4 pk;
            pk.x = pack_bf16x2(kv[0], kv[1]); pk.y = pack_bf16x2(kv[2], kv[3]);
            pk.z = pack_bf16x2(kv[4], kv[5]); pk.w = pack_bf16x2(kv[6], kv[7]);
            *reinterpret_cast<uint4 *>(&sK[key * LDK + part * 8]) = pk;
            // V: transposed, keys permuted inside each 32-block: position = kb*32 + gg*8 + t*4 + r for
            // key = kb*32 + t*16 + gg*4 + r
            const int kb = key >> 5, t = (key >> 4) & 1, gg = (key >> 2) & 3, r = key & 3;
            const int pos = kb * 32 + gg * 8 + t * 4 + r;
            const uint32_t vw[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sVt[(part * 8 + 2 * e) * LDV + pos] = (uint16_t)(vw[e] & 0xffffu);
                sVt[(part * 8 + 2 * e + 1) * LDV + pos] = (uint16_t)(vw[e] >> 16);
            }
        }
        __syncthreads();
        if (tile + 1 < ntiles) issue(tile + 1);  // next tile's loads fly during the MFMAs below

        // ---- S^T = K Q^T : s[kf][r] = S[key = kf*16 + g*4 + r][q = c16]
        f32x4 s[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            s[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 kfrag = *reinterpret_cast<const bf16x8 *>(&sK[(kf * 16 + c16) * LDK + kk * 32 + g * 8]);
                s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[kk], s[kf], 0, 0, 0);
            }
        }
        // mask the tail keys, running max
        float tmax = -1e30f;
        const int kbase = tile * KB + g * 4;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kbase + kf * 16 + r >= Lk) s[kf][r] = -1e30f;
                tmax = fmaxf(tmax, s[kf][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pf[2];
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
        for (int df = 0; df < 4; ++df) {
            o[df][0] *= alpha; o[df][1] *= alpha; o[df][2] *= alpha; o[df][3] *= alpha;
        }
        // ---- O^T += V^T P^T : o[df][r] = O[q = c16][d = df*16 + g*4 + r]
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 vfrag = *reinterpret_cast<const bf16x8 *>(&sVt[(df * 16 + c16) * LDV + kb * 32 + g * 8]);
                o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pf[kb], o[df], 0, 0, 0);
            }
    }

    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_run;
    const int qi = q0 + c16;
    if (qi < Lq) {
        uint16_t *op = a.out + ((size_t)b * Lq + qi) * a.out_stride + h * HD + g * 4;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const uint2 p = make_uint2(pack_bf16x2(o[df][0] * inv, o[df][1] * inv), pack_bf16x2(o[df][2] * inv, o[df][3] * inv));
            *reinterpret_cast<uint2 *>(op + df * 16) = p;
        }
    }
}

}  // namespace gadit

extern "C" int ga_attention_bf16(const GaAttentionArgs *a, void *stream)
{
    using namespace gadit;
    if (!a || !a->q || !a->k || !a->v || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->batch <= 0 || a->heads <= 0 || a->Lq <= 0 || a->Lk <= 0 || a->q_stride % 8 || a->k_stride % 8 ||
        a->v_stride % 8 || a->out_stride % 4)
        return GA_DIT_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((a->Lq + QB - 1) / QB, a->heads, a->batch);
    hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(256), 0, s, *a);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}
