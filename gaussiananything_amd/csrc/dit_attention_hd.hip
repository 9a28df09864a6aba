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

#include <type_traits>

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

// Round 6: the TUNED variant, taken when the caller passes V^T (GaAttentionHdArgs.vt: the image the projection GEMM's epilogue stores for
// any width, [batch * heads * head_dim][vt_ld], keys contiguous).  What changed against the kernel above:
//   * V^T arrives transposed: both tiles are 16-byte row copies (the 2-byte transposing LDS stores are gone);
//   * 128-query workgroups of EIGHT waves x 16 queries where they give the chip a round of work (a CFG pair's self-attention: 192): two
//     waves per SIMD overlap each other's LDS / exponent / MFMA phases; 64-query workgroups of four waves otherwise (one sample's
//     cross-attention: 192 of those) -- there with TWO KEY GROUPS of four waves (KS = 2: alternate tiles, buffers of their own, one merge
//     of (max, sum, O) through LDS at the end); four waves x 32 queries (QF = 2: every fragment read feeds two MFMAs) measured behind all of them;
//   * grid (heads * batch, query tiles): the workgroups of one (batch, head) sit on one XCD and share its L2 copy of K / V^T;
//   * two LDS buffers, tile t + 1 in registers while tile t is multiplied: one barrier per tile;
//   * the head dim in 16-wide steps, not 32: 72 = two MFMA k-steps of 32 and one of 16 for Q K^T (80 columns, not 96), five d tiles for P V;
//   * q's per-head RMSNorm here (q_norm_weight): the lane groups that hold a row's fragments add up its squares with two lane swaps -- a
//     launch per attention less; the softmax scale goes into the exponent's FMA instead of a second bf16 rounding of q; k's
//     (k_norm_weight) by the 4 / 8 neighbouring lanes that stage a key row, between the tile's arrival in registers and its LDS store.
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_hd __attribute__((ext_vector_type(4)));

// tail workgroups (dit_common.h: PrefetchJob -- weights of GEMMs a few launches ahead pulled towards the Infinity Cache by the CUs the
// attention grid leaves idle, as on the 64-wide path)
struct HdTail {
    PrefetchJob pf;
    int y0, nwgs;
    ShiftBiasJob job;     // sb_wgs > 0: the first sb_wgs tail workgroups compute it (the shift rows of the NEXT block's folded pre-norms)
    int sb_wgs;
};

template <int B_, int E_, class F>
__device__ __forceinline__ void static_for_hd(F &&f)
{
    if constexpr (B_ < E_) {
        f(std::integral_constant<int, B_>{});
        static_for_hd<B_ + 1, E_>(f);
    }
}

// KS = 2: two key groups of NW / 2 query waves each -- group ks takes the tiles ks, ks + 2, ... through buffers of its own, the groups' (max, sum,
// O) meet in LDS at the end: where even 64-query workgroups leave the SIMDs one wave each (one sample's attention: 192 workgroups)
template <int HD16, int QF, int NW = 4, int KS = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(1, 2))) void attention_hdv_kernel(GaAttentionHdArgs a, HdTail tail)
{
    constexpr int NTW = 64 * NW;            // threads of the workgroup
    if (tail.nwgs + tail.sb_wgs > 0 && (int)blockIdx.y >= tail.y0) {      // workgroups behind the attention grid (dispatched last: they land on the CUs it leaves idle)
        extern __shared__ __attribute__((aligned(16))) uint16_t smem_tail[];
        const int idx = ((int)blockIdx.y - tail.y0) * (int)gridDim.x + (int)blockIdx.x;
        if (idx < tail.sb_wgs) shift_bias_block<1>(tail.job, idx, reinterpret_cast<float *>(smem_tail));
        else if (idx - tail.sb_wgs < tail.nwgs) prefetch_block(tail.pf, idx - tail.sb_wgs, tail.nwgs);
        return;
    }
    constexpr int NWQ = NW / KS, NT = NTW / KS;     // query waves (QF x 16 queries each) and threads of one key group
    constexpr int KB = 64, HDP = HD16 * 16, KROW = HDP + 8, VROW = KB + 8, NK32 = HD16 / 2;
    constexpr bool K16 = (HD16 & 1) != 0;
    // K staging: TPR threads per key row (row tid / TPR), thread q of a row moves its 16-byte chunks q, q + TPR, ... -- a row's chunks sit in
    // TPR neighbouring lanes, so k's per-head RMSNorm (k_norm_weight) is a lane-local sum and log2(TPR) xor-shuffles at commit time
    constexpr int TPR = NT / KB, KCH = (HD16 * 2 + TPR - 1) / TPR, VCH = (HDP * 8 + NT - 1) / NT;   // chunks per thread and tile (upper bounds)
    constexpr int KT = KB * KROW, VT = HDP * VROW;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_hd[];     // K[2][key][d] (d >= head_dim zero), V^T[2][d][key] (rows >= head_dim zero)
    const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), g = lane >> 4, c16 = lane & 15;
    const int ks = wave_all / NWQ, wave = wave_all - ks * NWQ, tid = (int)threadIdx.x - ks * NT;     // (tid: inside the key group)
    uint16_t *Ks2 = smem_hd + ks * (2 * KT + 2 * VT), *Vt2 = Ks2 + 2 * KT;
    const int hd = a.head_dim, cpr = hd >> 3;
    const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads, q0 = blockIdx.y * (16 * NWQ * QF) + wave * (16 * QF);
    const int Lq = a.Lq, Lk = a.Lk;
    for (int i = threadIdx.x; i < KS * (2 * KT + 2 * VT) / 8; i += NTW) reinterpret_cast<uint4 *>(smem_hd)[i] = make_uint4(0u, 0u, 0u, 0u);

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[q][kk*32 + g*8 .. +7] (k-steps of 32) and Q[q][NK32*32 + g*4 .. +3] (the step of 16)
    bf16x8 qf[QF][NK32 > 0 ? NK32 : 1];
    bf16x4 qh[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int row = min(q0 + f * 16 + c16, Lq - 1);
        const uint16_t *qp = a.q + ((size_t)b * Lq + row) * a.q_stride + (size_t)h * hd;
        uint4 raw[NK32 > 0 ? NK32 : 1];
        uint2 rawh = make_uint2(0u, 0u);
#pragma unroll
        for (int kk = 0; kk < NK32; ++kk) {
            raw[kk] = make_uint4(0u, 0u, 0u, 0u);
            if (kk * 4 + g < cpr) raw[kk] = *reinterpret_cast<const uint4 *>(qp + kk * 32 + g * 8);
        }
        if (K16 && NK32 * 32 + g * 4 < hd) rawh = *reinterpret_cast<const uint2 *>(qp + NK32 * 32 + g * 4);
        if (a.q_norm_weight) {      // kernel-uniform
            float ss = 0.f;
#pragma unroll
            for (int kk = 0; kk < NK32; ++kk) {
                const uint32_t w[4] = {raw[kk].x, raw[kk].y, raw[kk].z, raw[kk].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                    ss += lo * lo + hi * hi;
                }
            }
            if (K16) {
                const uint32_t w[2] = {rawh.x, rawh.y};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
                    ss += lo * lo + hi * hi;
                }
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float rs = rsqrtf(ss / (float)hd + 1e-5f);
#pragma unroll
            for (int kk = 0; kk < NK32; ++kk) {
                if (kk * 4 + g < cpr) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(a.q_norm_weight + kk * 32 + g * 8);
                    const float4 w1 = *reinterpret_cast<const float4 *>(a.q_norm_weight + kk * 32 + g * 8 + 4);
                    const uint32_t w[4] = {raw[kk].x, raw[kk].y, raw[kk].z, raw[kk].w};
                    const float ws[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                    uint32_t o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o4[e] = pack_bf16x2(__uint_as_float(w[e] << 16) * (rs * ws[2 * e]), __uint_as_float(w[e] & 0xffff0000u) * (rs * ws[2 * e + 1]));
                    raw[kk] = make_uint4(o4[0], o4[1], o4[2], o4[3]);
                }
            }
            if (K16 && NK32 * 32 + g * 4 < hd) {
                const float4 w0 = *reinterpret_cast<const float4 *>(a.q_norm_weight + NK32 * 32 + g * 4);
                rawh = make_uint2(pack_bf16x2(__uint_as_float(rawh.x << 16) * (rs * w0.x), __uint_as_float(rawh.x & 0xffff0000u) * (rs * w0.y)),
                                  pack_bf16x2(__uint_as_float(rawh.y << 16) * (rs * w0.z), __uint_as_float(rawh.y & 0xffff0000u) * (rs * w0.w)));
            }
        }
#pragma unroll
        for (int kk = 0; kk < NK32; ++kk) qf[f][kk] = __builtin_bit_cast(bf16x8, raw[kk]);
        qh[f] = __builtin_bit_cast(bf16x4, rawh);
    }
    const float cs = rsqrtf((float)hd) * 1.4426950408889634f;   // softmax scale x log2(e): applied inside the exponent's FMA

    f32x4 o[QF][HD16];
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        m_run[f] = -1e30f; l_run[f] = 0.f;
#pragma unroll
        for (int i = 0; i < HD16; ++i) o[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const uint16_t *kb_ = a.k + (size_t)b * Lk * a.k_stride + (size_t)h * hd;
    const uint16_t *vb_ = a.vt + ((size_t)b * a.heads + h) * hd * a.vt_ld;
    const int ntiles = (Lk + KB - 1) / KB;
    // this thread's chunks of a tile: K chunks (row tid / TPR, part tid % TPR + TPR i); V^T chunk c = tid + NT i -> (d row c >> 3, part c & 7)
    // (compile-time indices throughout -- static_for, not unrolled loops inside the lambdas: those left the staging registers in scratch)
    int v_lds[VCH];
    const uint16_t *v_src[VCH];
    const int k_row = tid / TPR, k_q = tid - k_row * TPR;
    bool k_ok[KCH];
    static_for_hd<0, KCH>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; k_ok[i] = k_q + TPR * i < cpr; });
    // k's norm weights of this thread's chunks (zero where there is no chunk, or no norm)
    float kw[KCH][8];
    static_for_hd<0, KCH>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
        if (a.k_norm_weight && k_ok[i]) {
            w0 = *reinterpret_cast<const float4 *>(a.k_norm_weight + (k_q + TPR * i) * 8);
            w1 = *reinterpret_cast<const float4 *>(a.k_norm_weight + (k_q + TPR * i) * 8 + 4);
        }
        kw[i][0] = w0.x; kw[i][1] = w0.y; kw[i][2] = w0.z; kw[i][3] = w0.w; kw[i][4] = w1.x; kw[i][5] = w1.y; kw[i][6] = w1.z; kw[i][7] = w1.w;
    });
    static_for_hd<0, VCH>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const int c = tid + NT * i, row = c >> 3, part = c & 7;
        v_src[i] = vb_ + (size_t)min(row, hd - 1) * a.vt_ld + part * 8;
        v_lds[i] = row < hd ? row * VROW + part * 8 : -1;
    });
    u32x4_hd kreg[KCH], vreg[VCH];     // (LLVM vector values, not HIP_vector_type structs: those were copied with memcpy and stayed in scratch)
    auto issue = [&](int t) __attribute__((always_inline)) {
        static_for_hd<0, KCH>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            // (no branch around a load: a chunk slot past the tile re-reads a valid address and is not committed.  With branches the register
            //  allocator joins the paths through copies of the loaded registers -- s_waitcnt vmcnt(0) straight after the barrier)
            kreg[i] = *reinterpret_cast<const u32x4_hd *>(kb_ + (size_t)min(t * KB + k_row, Lk - 1) * a.k_stride + (k_ok[i] ? k_q + TPR * i : 0) * 8);
        });
        static_for_hd<0, VCH>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            vreg[i] = *reinterpret_cast<const u32x4_hd *>(v_src[i] + t * KB);
        });
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        if (a.k_norm_weight) {      // kernel-uniform: k rows RMS-normalised on their way into LDS (bf16-rounded like ga_head_rmsnorm_bf16's)
            float ss = 0.f;
            static_for_hd<0, KCH>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(kreg[i][e] << 16), hi = __uint_as_float(kreg[i][e] & 0xffff0000u);
                    ss += k_ok[i] ? lo * lo + hi * hi : 0.f;
                }
            });
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) ss += __shfl_xor(ss, o, 64);
            const float rs = rsqrtf(ss / (float)hd + 1e-5f);
            static_for_hd<0, KCH>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    kreg[i][e] = pack_bf16x2(__uint_as_float(kreg[i][e] << 16) * (rs * kw[i][2 * e]), __uint_as_float(kreg[i][e] & 0xffff0000u) * (rs * kw[i][2 * e + 1]));
            });
        }
        static_for_hd<0, KCH>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if (k_ok[i]) *reinterpret_cast<u32x4_hd *>(Ks2 + buf * KT + k_row * KROW + (k_q + TPR * i) * 8) = kreg[i];
        });
        static_for_hd<0, VCH>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if (v_lds[i] >= 0) *reinterpret_cast<u32x4_hd *>(Vt2 + buf * VT + v_lds[i]) = vreg[i];
        });
    };
    issue(min(ks, ntiles - 1));
    __syncthreads();                               // the zero fill is done
    commit(0);
    const int krow = 8 * (c16 >> 2) + (c16 & 3);
    const int nit = (ntiles + KS - 1) / KS;
    for (int it = 0; it < nit; ++it) {
        const int t = it * KS + ks;                // this key group's tile of the iteration (KS = 2, odd tile count: the last one may not exist)
        issue(min(t + KS, ntiles - 1));            // in flight while tile t is multiplied (past the end: a harmless re-fetch -- unconditional, so
                                                   // that the staging registers are plain values, not merged paths)
        // tile t is committed by everybody; everybody has left tile t - 1 (the other buffer).  NOT __syncthreads(): its fence waits for
        // vmcnt(0) -- loads and stores share that counter on gfx9 -- i.e. for the tile just requested: every global round trip back on the
        // critical path (the first version of this kernel, and round 6's double-buffering of the kernel above, measured exactly that)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const uint16_t *Ks = Ks2 + (it & 1) * KT, *Vt = Vt2 + (it & 1) * VT;
        if (KS == 1 || t < ntiles) {               // wave-uniform
        // S^T = K Q^T : s[f][kf][r] <-> key 32 (kf >> 1) + 8 g + 4 (kf & 1) + r of the tile, query c16 of fragment f
        f32x4 s[QF][4];
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) s[f][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < NK32; ++kk)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const bf16x8 kfrag = *reinterpret_cast<const bf16x8 *>(Ks + ((kf >> 1) * 32 + (kf & 1) * 4 + krow) * KROW + kk * 32 + g * 8);
#pragma unroll
                for (int f = 0; f < QF; ++f) s[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, qf[f][kk], s[f][kf], 0, 0, 0);
            }
        if (K16) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const bf16x4 kfrag = *reinterpret_cast<const bf16x4 *>(Ks + ((kf >> 1) * 32 + (kf & 1) * 4 + krow) * KROW + NK32 * 32 + g * 4);
#pragma unroll
                for (int f = 0; f < QF; ++f) s[f][kf] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kfrag, qh[f], s[f][kf], 0, 0, 0);
            }
        }
        const int kbase = t * KB + g * 8;
        const bool edge = (t + 1) * KB > Lk;       // uniform: only the last tile masks
        bf16x8 pf[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float tmax = -1e30f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (edge && kbase + (kf >> 1) * 32 + (kf & 1) * 4 + r >= Lk) s[f][kf][r] = -1e30f;
                    tmax = fmaxf(tmax, s[f][kf][r]);
                }
            tmax = group_max_hd(tmax);
            const float m_new = fmaxf(m_run[f], tmax * cs);
            const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
            m_run[f] = m_new;
            float psum = 0.f;
            uint32_t pw[2][4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(s[f][kf][0], cs, -m_new)), p1 = __builtin_amdgcn_exp2f(fmaf(s[f][kf][1], cs, -m_new));
                const float p2 = __builtin_amdgcn_exp2f(fmaf(s[f][kf][2], cs, -m_new)), p3 = __builtin_amdgcn_exp2f(fmaf(s[f][kf][3], cs, -m_new));
                psum += (p0 + p1) + (p2 + p3);
                pw[kf >> 1][(kf & 1) * 2] = pack_bf16x2(p0, p1);
                pw[kf >> 1][(kf & 1) * 2 + 1] = pack_bf16x2(p2, p3);
            }
            pf[f][0] = __builtin_bit_cast(bf16x8, make_uint4(pw[0][0], pw[0][1], pw[0][2], pw[0][3]));
            pf[f][1] = __builtin_bit_cast(bf16x8, make_uint4(pw[1][0], pw[1][1], pw[1][2], pw[1][3]));
            l_run[f] = l_run[f] * alpha + psum;
#pragma unroll
            for (int df = 0; df < HD16; ++df) { o[f][df][0] *= alpha; o[f][df][1] *= alpha; o[f][df][2] *= alpha; o[f][df][3] *= alpha; }
        }
        // O^T += V^T P^T : o[f][df][r] = O[q = c16][d = df*16 + g*4 + r]; the lane's 8 P of block kb are keys 32 kb + 8 g ..
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int df = 0; df < HD16; ++df) {
                const bf16x8 vfrag = *reinterpret_cast<const bf16x8 *>(Vt + (df * 16 + c16) * VROW + kb * 32 + g * 8);
#pragma unroll
                for (int f = 0; f < QF; ++f) o[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag, pf[f][kb], o[f][df], 0, 0, 0);
            }
        }
        commit((it + 1) & 1);                      // (its last readers passed this iteration's barrier an iteration ago; after the last tile nobody reads it)
    }
    if constexpr (KS == 2) {
        // the second key group hands its (max, lane-partial sum, O) over through LDS; the first merges and stores
        constexpr int MW = QF * (2 + 4 * HD16);    // floats per lane
        static_assert(NWQ * 64 * MW * 4 <= (2 * KT + 2 * VT) * 2, "the hand-over fits the first group's buffers");
        __syncthreads();                           // everybody has left the tiles
        float *mg = reinterpret_cast<float *>(smem_hd) + (size_t)(wave * 64 + lane) * MW;
        if (ks == 1) {
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                mg[f * (2 + 4 * HD16)] = m_run[f]; mg[f * (2 + 4 * HD16) + 1] = l_run[f];
#pragma unroll
                for (int df = 0; df < HD16; ++df)
                    *reinterpret_cast<float4 *>(mg + f * (2 + 4 * HD16) + 2 + 4 * df) = make_float4(o[f][df][0], o[f][df][1], o[f][df][2], o[f][df][3]);
            }
        }
        __syncthreads();
        if (ks == 1) return;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            const float m1 = mg[f * (2 + 4 * HD16)], l1 = mg[f * (2 + 4 * HD16) + 1];
            const float m = fmaxf(m_run[f], m1);
            const float a0 = __builtin_amdgcn_exp2f(m_run[f] - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
            l_run[f] = l_run[f] * a0 + l1 * a1;
#pragma unroll
            for (int df = 0; df < HD16; ++df) {
                const float4 o1 = *reinterpret_cast<const float4 *>(mg + f * (2 + 4 * HD16) + 2 + 4 * df);
                o[f][df][0] = o[f][df][0] * a0 + o1.x * a1; o[f][df][1] = o[f][df][1] * a0 + o1.y * a1;
                o[f][df][2] = o[f][df][2] * a0 + o1.z * a1; o[f][df][3] = o[f][df][3] * a0 + o1.w * a1;
            }
        }
    }
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        const int row = q0 + f * 16 + c16;
        if (row < Lq) {
            uint16_t *op = a.out + ((size_t)b * Lq + row) * a.out_stride + (size_t)h * hd + g * 4;
#pragma unroll
            for (int df = 0; df < HD16; ++df)
                if (df * 16 + g * 4 < hd)
                    *reinterpret_cast<uint2 *>(op + df * 16) = make_uint2(pack_bf16x2(o[f][df][0] * inv, o[f][df][1] * inv), pack_bf16x2(o[f][df][2] * inv, o[f][df][3] * inv));
        }
    }
}

template <int HD16>
static int launch_hdv(const GaAttentionHdArgs &a, hipStream_t s, const PrefetchJob *pf, int pf_wgs, const ShiftBiasJob *job)
{
    static_assert((size_t)(2 * 64 * (HD16 * 16 + 8) + 2 * HD16 * 16 * 72) * 2 >= kSbLdsFloats * sizeof(float) || HD16 < 2, "the tail's partial sums fit the tile buffers");
    constexpr size_t lds = (size_t)(2 * 64 * (HD16 * 16 + 8) + 2 * HD16 * 16 * 72) * 2;
    // configurations: 1 = 4 waves x 16 queries (64-query workgroups), 2 = 4 waves x 32 (128), 3 = 8 waves x 16 (128: two waves per SIMD
    // hide each other's LDS / exponent latencies), 4 = two key groups of 4 waves x 16 (64).  GA_ATTN_HD_QF forces one (A/B aid)
    static const int qf_env = [] { const char *e = getenv("GA_ATTN_HD_QF"); return e ? atoi(e) : 0; }();
    const long long wg128 = (long long)a.batch * a.heads * ((a.Lq + 127) / 128);
    // same-box (tools/attn_hd_bench.py, 16 heads of 72; us for configurations 1 | 2 | 3 | 4): a CFG pair's self-attention 23.4 | 25.2 | 21.4 | 27.2,
    // one sample's cross-attention 27.9 | 40.3 | 33.6 | 21.6, one sample's self-attention 17.5 | 24.7 | 20.5 | 14.6 -- one wave per SIMD runs its
    // phases back to back, two overlap; below a round of 128-query workgroups the 64-query ones fill more CUs, and two key groups give each
    // SIMD its second wave and halve the tiles a wave walks
    const int cfg = qf_env ? qf_env : (wg128 >= 160 ? 3 : 4);
    const int qpw = (cfg == 1 || cfg == 4) ? 64 : 128;
    dim3 grid((unsigned)(a.batch * a.heads), (unsigned)((a.Lq + qpw - 1) / qpw));
    HdTail tail{};
    tail.y0 = (int)grid.y;
    if (pf && pf_wgs > 0) { tail.pf = *pf; tail.nwgs = pf_wgs; }
    if (job) {
        // the job's waves each own K / 64 / waves K-tiles, at most kSbTilesPerWave (dit_common.h); its partial sums live in the tile buffers
        const int waves = cfg == 3 || cfg == 4 ? 8 : 4;
        if (job->K / 64 > kSbTilesPerWave * waves || lds < kSbLdsFloats * sizeof(float)) return GA_DIT_ERR_BAD_SHAPE;
        tail.job = *job; tail.sb_wgs = shift_bias_wgs(job->N0, job->N1);
    }
    if (tail.nwgs + tail.sb_wgs > 0) grid.y += (unsigned)((tail.nwgs + tail.sb_wgs + (int)grid.x - 1) / (int)grid.x);
#define GA_HDV_LAUNCH(QFV, NWV, KSV)                                                                                              \
    do {                                                                                                                          \
        if (KSV * lds > 65536 && hipFuncSetAttribute((const void *)attention_hdv_kernel<HD16, QFV, NWV, KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KSV * lds)) != hipSuccess) return GA_DIT_ERR_LAUNCH; \
        hipLaunchKernelGGL((attention_hdv_kernel<HD16, QFV, NWV, KSV>), grid, dim3(64 * NWV), KSV * lds, s, a, tail);              \
    } while (0)
    if (cfg == 2) GA_HDV_LAUNCH(2, 4, 1);
    else if (cfg == 3) GA_HDV_LAUNCH(1, 8, 1);
    else if (cfg == 4) GA_HDV_LAUNCH(1, 8, 2);
    else GA_HDV_LAUNCH(1, 4, 1);
#undef GA_HDV_LAUNCH
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
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

namespace gadit {
static int attention_hd_dispatch(const GaAttentionHdArgs *a, void *stream, const PrefetchJob *pf, int pf_wgs, const ShiftBiasJob *job);
// the V^T variant with tail workgroups (ga_dit_forward); attention_hd_workgroups: the grid without them
int attention_hd_with_tail(const GaAttentionHdArgs *a, void *stream, const PrefetchJob *pf, int pf_wgs, const ShiftBiasJob *job)
{
    return attention_hd_dispatch(a, stream, pf, pf_wgs, job);
}
// can a tail of this launch host a ShiftBiasJob over K columns?  (eight-wave configurations only: the launcher's own choice unless forced)
bool attention_hd_hosts_shift_bias(const GaAttentionHdArgs *a, int K)
{
    static const int qf_env = [] { const char *e = getenv("GA_ATTN_HD_QF"); return e ? atoi(e) : 0; }();
    const int waves = (qf_env == 1 || qf_env == 2) ? 4 : 8;
    const size_t lds = (size_t)(2 * 64 * ((a->head_dim + 15) / 16 * 16 + 8) + 2 * ((a->head_dim + 15) / 16 * 16) * 72) * 2;
    return K / 64 <= kSbTilesPerWave * waves && lds >= kSbLdsFloats * sizeof(float);
}
int attention_hd_workgroups(const GaAttentionHdArgs *a)
{
    const long long wg128 = (long long)a->batch * a->heads * ((a->Lq + 127) / 128);
    return (int)(wg128 >= 160 ? wg128 : (long long)a->batch * a->heads * ((a->Lq + 63) / 64));
}
}  // namespace gadit

extern "C" int ga_attention_hd_bf16(const GaAttentionHdArgs *a, void *stream) { return gadit::attention_hd_dispatch(a, stream, nullptr, 0, nullptr); }

static int gadit::attention_hd_dispatch(const GaAttentionHdArgs *a, void *stream, const PrefetchJob *pf, int pf_wgs, const ShiftBiasJob *job)
{
    using namespace gadit;
    if (!a || !a->q || !a->k || (!a->v && !a->vt) || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->batch <= 0 || a->heads <= 0 || a->Lq <= 0 || a->Lk <= 0 || a->head_dim < 8 || a->head_dim > 128 || a->head_dim % 8 ||
        a->q_stride % 8 || a->k_stride % 8 || a->out_stride % 4)
        return GA_DIT_ERR_BAD_SHAPE;
    if (((uintptr_t)a->q | (uintptr_t)a->k) % 16 != 0 || (uintptr_t)a->out % 8 != 0) return GA_DIT_ERR_BAD_SHAPE;
    if (a->vt) {     // the tuned variant: V^T [batch * heads * head_dim][vt_ld], vt_ld a multiple of 64 that covers the keys
        if (a->vt_ld % 64 != 0 || a->vt_ld < (a->Lk + 63) / 64 * 64 || (uintptr_t)a->vt % 16 != 0 || (a->q_norm_weight && (uintptr_t)a->q_norm_weight % 16 != 0) ||
            (a->k_norm_weight && (uintptr_t)a->k_norm_weight % 16 != 0))
            return GA_DIT_ERR_BAD_SHAPE;
        hipStream_t sv = reinterpret_cast<hipStream_t>(stream);
        switch ((a->head_dim + 15) / 16) {
        case 1: return launch_hdv<1>(*a, sv, pf, pf_wgs, job);
        case 2: return launch_hdv<2>(*a, sv, pf, pf_wgs, job);
        case 3: return launch_hdv<3>(*a, sv, pf, pf_wgs, job);
        case 4: return launch_hdv<4>(*a, sv, pf, pf_wgs, job);
        case 5: return launch_hdv<5>(*a, sv, pf, pf_wgs, job);
        case 6: return launch_hdv<6>(*a, sv, pf, pf_wgs, job);
        case 7: return launch_hdv<7>(*a, sv, pf, pf_wgs, job);
        default: return launch_hdv<8>(*a, sv, pf, pf_wgs, job);
        }
    }
    if (a->q_norm_weight || a->k_norm_weight) return GA_DIT_ERR_BAD_SHAPE;     // (the norms inside the kernel belong to the V^T variant)
    if (a->v_stride % 8 || (uintptr_t)a->v % 16 != 0) return GA_DIT_ERR_BAD_SHAPE;
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
