// dit_attention.hip -- fused multi-head attention forward (no mask, no dropout), head_dim 64, bf16 MFMA, gfx950.
//
// Computes what the reference obtains from xformers.ops.memory_efficient_attention between the projections of
//   MemEffAttention.forward               /root/reference/vit/vision_transformer.py:284-297  (self-attention, L = 768)
//   MemoryEfficientCrossAttention.forward /root/reference/ldm/modules/attention.py:514-548   (image tokens, Lk = 1369)
// INCLUDING the per-head RMSNorm of q and k that precedes it (learned weight[64], eps 1e-5, dit/norm.py:29-43), which
// is applied while the tiles are staged, and the softmax scale 64^-1/2.
//
// MI355X mapping (flash-style, one pass over the keys, online softmax in fp32):
//   * grid (ceil(Lq/128), heads, batch); 8 waves per workgroup, each wave owns 16 query rows; two waves share a SIMD
//     so that one wave's softmax (VALU) overlaps the other's MFMAs; the workgroup shares the staged K / V^T tiles
//     (64 keys) in LDS;
//   * "swapped" products so that every reduction is lane-local or a 2-step lane shuffle: S^T = K Q^T puts one query
//     in a lane (column lane&15) with 4 keys per accumulator fragment, and O^T = V^T P^T keeps that query in the same
//     lane, so the running max / sum / rescale never cross lanes except for two xor-shuffles per tile;
//   * P never leaves registers: the PV product defines its own key order inside each 32-key block (lane-group g,
//     element e  <->  key 16*(e>>2) + 4*g + (e&3)), which is exactly how the S^T accumulators already sit in the lane;
//     V arrives TRANSPOSED from the projection GEMM's epilogue (keys contiguous), so staging it is two 8-byte LDS
//     writes per 16-byte chunk (the permutation above) and its MFMA fragments are plain 16-byte LDS reads;
//   * tiles are double-buffered in LDS (global -> VGPR one tile ahead, VGPR -> LDS for tile t+1 before the math of
//     tile t), one barrier per tile; rows are 128 bytes with the 16-byte slot XOR-swizzled by (row & 7): conflict-free
//     for the hardware's ds_read_b128 lane groups;
//   * K rows are RMS-normalised by the 8 lanes that stage a row (3 xor-shuffles), Q rows by the 4 lane-groups that
//     hold a row's fragments; the softmax scale and log2(e) are folded into Q so the exponentials are bare v_exp_f32.
#include "dit_common.h"

namespace gadit {

constexpr int KB = 64, HD = 64;
constexpr int TILE = KB * HD;  // elements of one staged tile (8 KiB)

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ (row & 7)) * 8); }

// NW waves per workgroup, QI 16-row query fragments per wave: query rows per workgroup = NW * QI * 16.
// <8,1>: 8 waves x 16 rows -- two waves per SIMD, so one wave's softmax VALU work overlaps the other's MFMAs (PMC of the
// <4,2> shape: one wave per SIMD, 58 % of its cycles spent issuing ~9 k instructions at the lone-wave rate of one per
// ~4.5 cycles, MFMA pipe 7 % busy).
template <int NW, int QI>
__global__ __launch_bounds__(NW * 64) void attention_fwd_kernel(GaAttentionArgs a)
{
    constexpr int QB = NW * QI * 16, NT = NW * 64, CPT = 512 / NT;  // chunks (16 B) per thread per staged tile
    __shared__ __attribute__((aligned(16))) uint16_t sK[2 * TILE];   // [buf][key][d]      (swizzled)
    __shared__ __attribute__((aligned(16))) uint16_t sV[2 * TILE];   // [buf][d][perm key] (swizzled)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wave * (QI * 16);
    const int Lq = a.Lq, Lk = a.Lk;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + qi*16 + c16][kk*32 + g*8 .. +7], normalised, scaled
    bf16x8 qf[QI][2];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        const int row = min(q0 + qi * 16 + c16, Lq - 1);
        const uint16_t *qp = a.q + ((size_t)b * Lq + row) * a.q_stride + h * HD;
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
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = qv[kk * 8 + e];
                if (a.q_norm_weight) v *= a.q_norm_weight[kk * 32 + g * 8 + e];
                qf[qi][kk][e] = (short)f32_to_bf16(v * rs);
            }
    }

    f32x4 o[QI][4];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[qi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QI], l_run[QI];
#pragma unroll
    for (int qi = 0; qi < QI; ++qi) { m_run[qi] = -1e30f; l_run[qi] = 0.f; }

    // ---- staging: K and V^T tiles are 512 16-byte chunks each, 2 per thread; chunk c -> row c>>3, part c&7
    const int ntiles = (Lk + KB - 1) / KB;
    const uint16_t *vt_base = a.vt + ((size_t)b * a.heads + h) * HD * a.vt_ld;
    uint4 rk[CPT], rv[CPT];
    auto issue = [&](int tile) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + NT * i, row = c >> 3, part = c & 7;
            const int key = min(tile * KB + row, Lk - 1);
            rk[i] = *reinterpret_cast<const uint4 *>(a.k + ((size_t)b * Lk + key) * a.k_stride + h * HD + part * 8);
            rv[i] = *reinterpret_cast<const uint4 *>(vt_base + (size_t)row * a.vt_ld + tile * KB + part * 8);
        }
    };
    auto write_lds = [&](int buf) {
        uint16_t *dk = sK + buf * TILE, *dv = sV + buf * TILE;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + NT * i, row = c >> 3, part = c & 7;
            // K: RMS-normalise the row (8 consecutive lanes hold it)
            const uint32_t kw[4] = {rk[i].x, rk[i].y, rk[i].z, rk[i].w};
            uint4 pk = rk[i];
            if (a.k_norm_weight) {
                float kv[8];
                float ss = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kv[2 * e] = __uint_as_float(kw[e] << 16);
                    kv[2 * e + 1] = __uint_as_float(kw[e] & 0xffff0000u);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += kv[e] * kv[e];
                ss += __shfl_xor(ss, 1, 64);
                ss += __shfl_xor(ss, 2, 64);
                ss += __shfl_xor(ss, 4, 64);
                const float rs = rsqrtf(ss * (1.0f / HD) + 1e-5f);
#pragma unroll
                for (int e = 0; e < 8; ++e) kv[e] *= rs * a.k_norm_weight[part * 8 + e];
                pk.x = pack_bf16x2(kv[0], kv[1]); pk.y = pack_bf16x2(kv[2], kv[3]);
                pk.z = pack_bf16x2(kv[4], kv[5]); pk.w = pack_bf16x2(kv[6], kv[7]);
            }
            *reinterpret_cast<uint4 *>(dk + swz(row, part)) = pk;
            // V^T row d = row, keys 8*part .. +7 of the tile: permute inside the 32-key block (see header)
            const int kb = part >> 2, cc = part & 3;
            const int pc0 = kb * 4 + 2 * (cc & 1), off = (cc >> 1) * 4;
            *reinterpret_cast<uint2 *>(dv + swz(row, pc0) + off) = make_uint2(rv[i].x, rv[i].y);
            *reinterpret_cast<uint2 *>(dv + swz(row, pc0 + 1) + off) = make_uint2(rv[i].z, rv[i].w);
        }
    };

    issue(0);
    write_lds(0);
    if (ntiles > 1) issue(1);
    __syncthreads();

    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        if (tile + 1 < ntiles) write_lds(cur ^ 1);     // buffer cur^1 was last read before the previous barrier
        if (tile + 2 < ntiles) issue(tile + 2);        // flies during the MFMAs below
        const uint16_t *bk = sK + cur * TILE, *bv = sV + cur * TILE;

        // ---- S^T = K Q^T : s[qi][kf][r] = S[key = kf*16 + g*4 + r][q = qi*16 + c16]
        f32x4 s[QI][4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
            for (int qi = 0; qi < QI; ++qi) s[qi][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 kfrag = *reinterpret_cast<const bf16x8 *>(bk + swz(kf * 16 + c16, kk * 4 + g));
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
                    s[qi][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[qi][kk], s[qi][kf], 0, 0, 0);
            }
        }
        const int kbase = tile * KB + g * 4;
        const bool tail = tile * KB + KB > Lk;
        bf16x8 pf[QI][2];
#pragma unroll
        for (int qi = 0; qi < QI; ++qi) {
            float tmax = -1e30f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (tail && kbase + kf * 16 + r >= Lk) s[qi][kf][r] = -1e30f;
                    tmax = fmaxf(tmax, s[qi][kf][r]);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run[qi], tmax);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qi] - m_new);
            m_run[qi] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[qi][kf][r] - m_new);
                    psum += p;
                    pf[qi][kf >> 1][(kf & 1) * 4 + r] = (short)f32_to_bf16(p);
                }
            l_run[qi] = l_run[qi] * alpha + psum;
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                o[qi][df][0] *= alpha; o[qi][df][1] *= alpha; o[qi][df][2] *= alpha; o[qi][df][3] *= alpha;
            }
        }
        // ---- O^T += V^T P^T : o[qi][df][r] = O[q = qi*16 + c16][d = df*16 + g*4 + r]
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 vfrag = *reinterpret_cast<const bf16x8 *>(bv + swz(df * 16 + c16, kb * 4 + g));
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
                    o[qi][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pf[qi][kb], o[qi][df], 0, 0, 0);
            }
        __syncthreads();
    }

#pragma unroll
    for (int qi = 0; qi < QI; ++qi) {
        float l = l_run[qi];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int row = q0 + qi * 16 + c16;
        if (row < Lq) {
            uint16_t *op = a.out + ((size_t)b * Lq + row) * a.out_stride + h * HD + g * 4;
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                const uint2 p = make_uint2(pack_bf16x2(o[qi][df][0] * inv, o[qi][df][1] * inv),
                                           pack_bf16x2(o[qi][df][2] * inv, o[qi][df][3] * inv));
                *reinterpret_cast<uint2 *>(op + df * 16) = p;
            }
        }
    }
}

}  // namespace gadit

extern "C" int ga_attention_bf16(const GaAttentionArgs *a, void *stream)
{
    using namespace gadit;
    if (!a || !a->q || !a->k || !a->vt || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->batch <= 0 || a->heads <= 0 || a->Lq <= 0 || a->Lk <= 0 || a->q_stride % 8 || a->k_stride % 8 || a->vt_ld % 8 ||
        a->vt_ld < ((a->Lk + KB - 1) / KB) * KB || a->out_stride % 4)
        return GA_DIT_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    constexpr int QB = 128;  // 8 waves x 16 rows
    const dim3 grid((a->Lq + QB - 1) / QB, a->heads, a->batch);
    hipLaunchKernelGGL((attention_fwd_kernel<8, 1>), grid, dim3(512), 0, s, *a);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}
