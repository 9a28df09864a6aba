// dit_gemm.hip -- bf16 MFMA GEMM with the fused epilogues of the DiT block, gfx950.
//
//   acc[m][n] = sum_k A[m][k] * W[n][k]     A: activations [M,K] bf16, W: nn.Linear weight [N,K] bf16, fp32 accumulate
// Epilogues (include/ga_dit.h): bias + bf16 store (QKV / q / K|V projections), bias + erf-GELU + bf16 store (FusedMLP
// fc1: /root/reference/dit/dit_models_xformers.py:281-286), gated residual accumulate into the fp32 stream
// (x += gate * (acc + bias): dit_models_xformers.py:775-785), fp32 store.  Fusing them removes one full read+write
// of the [M,N] tensor per GEMM, which at M = 1536 rows is comparable to the GEMM's own operand traffic.
//
// MI355X mapping: both operands are K-contiguous, which is exactly the lane layout v_mfma_f32_16x16x32_bf16 wants for
// its A and B operands (lane l: row l&15, k-chunk (l>>4)*8..+7), so the same ds_read_b128 fragment load serves both.
// The MFMA "A" role is given to the WEIGHT rows and the "B" role to the activation rows: the accumulator fragment of a
// lane is then 4 consecutive n for one m, i.e. 8-byte (bf16) / 16-byte (fp32) contiguous stores in the row-major
// output.  Workgroup tile 128(n) x 128(m) x 64(k), 4 waves as 2x2, each wave 4x4 fragments (64 accumulator VGPRs).
// Staging is LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction straight into LDS, no VGPR round trip) through
// a ring of 4 slots (128 KiB of the CU's 160 KiB; 2 slots for the larger grids): K-tiles t+1..t+3 are in flight while tile t is multiplied, with COUNTED
// waits (s_waitcnt vmcnt(24/16/8/0): 8 DMA instructions per wave per tile) and a raw s_barrier -- one barrier per
// K-tile.  At M = 1536 the grid is only 96-384 workgroups, so every workgroup must run at MFMA speed on its own: with a
// 2-deep ring each K-tile cost a full DMA latency (measured 1.1 us vs 0.22 us of MFMA work).  The DMA writes LDS in
// lane order, so the bank-conflict-free image is obtained by permuting the SOURCE: slot (row, s) of the 128-byte row
// holds global 16-byte chunk s ^ (row & 7), and fragment reads apply the same XOR (conflict-free for the hardware's
// ds_read_b128 lane groups).
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "dit_common.h"

namespace gadit {

#ifndef GA_GEMM_ABLATE
#define GA_GEMM_ABLATE 0   // tools/gemm_ablate.sh: timing-only builds with phases removed (wrong results); bit mask:
                           // 1 MFMAs, 2 DMA in the K loop, 4 barrier, 8 epilogue, 16 fragment reads; 32 = no K loop at all
#endif
__device__ __attribute__((aligned(16))) const float g_zero_line[4] = {0.f, 0.f, 0.f, 0.f};   // stands in for an absent epilogue operand
constexpr int BN = 128, BK = 64;
constexpr int TILE_ELEMS = 128 * BK;  // one operand tile: 128 rows x 64 bf16 = 16 KiB

struct GemmP {  // by-value kernel parameters (kept flat: no pointer into the argument struct is taken)
    int M, N, K, rows_per_batch;
    const uint16_t *A, *W;
    const float *bias, *gate;
    void *out;
    long long lda, ldo, gate_stride;
    uint16_t *vt;      // optional transposed store of the columns >= vt_col0 (see include/ga_dit.h)
    int vt_col0, heads;
    long long vt_ld;
    const float *qk_w0, *qk_w1;  // optional per-head RMSNorm of the leading column groups
    int qk_cols0, qk_cols1;
    // folded un-modulated RMSNorm (include/ga_dit.h): producer side (EPI_RESIDUAL) / consumer side (EPI_STORE_BF16)
    uint16_t *emit_x;
    float *emit_ss;
    long long emit_ld;
    const float *row_ss;
    int row_ss_tiles;
    float row_ss_inv_dim, row_ss_eps;
    int w_tiled;       // W is the tiled image [N/8][K/64][8][64] (include/ga_dit.h)
    // folded MODULATED RMSNorm (include/ga_dit.h): emit multiplier w[n] (1 + scale_b[n]); one bias row per batch item; rows of A in the product
    const float *emit_w, *emit_scale;
    long long emit_scale_stride, bias_stride;
    int k_rows;
    // deterministic split-K (round 6, include/ga_dit.h: GaGemmArgs.splitk_ws): blockIdx.z takes K / splits of the reduction; the partial
    // tiles meet in sk_part, the last workgroup of a tile to arrive (sk_count) adds them in split order and runs the epilogue
    int splits;
    float *sk_part;
    unsigned *sk_count;
    // round 6: tile -> XCD blocking (ring kernels).  Workgroup ids are dealt to the 8 XCDs round-robin; with xmap the XCD c = id % 8 owns a
    // (gridDim.x / 2) x (gridDim.y / 4) block of output tiles instead of every 8th column of tiles: for the N = 1024 GEMMs, whose
    // activation operand is larger than the weight, an XCD's L2 then serves 7 of 8 reads of an activation tile instead of 1 of 2
    int xmap;
    int wt;      // write-through output stores (st16)
};

// Element offset of chunk c (8 bf16) of weight row n at K-tile 0, and the stride from one K-tile to the next: row-major
// [N][K] or the tiled image [N/8][K/64][8][64], in which the 8 rows x 128 bytes one LDS-DMA instruction moves are 1 KiB
// contiguous.
__device__ __forceinline__ size_t w_offset(const GemmP &p, int n, int c)
{
    return p.w_tiled ? ((size_t)(n >> 3) * (p.K >> 6) * 512 + (n & 7) * 64 + c * 8) : ((size_t)n * p.K + c * 8);
}
__device__ __forceinline__ int w_kstep(const GemmP &p) { return p.w_tiled ? 512 : BK; }

__device__ __forceinline__ void glds16(const uint16_t *gsrc, uint16_t *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// Round 6 (GemmP.wt, GA_GEMM_WT): output stores as agent-scope write-through (sc1) -- the tile's rows leave for the memory side while the
// other workgroups still compute, and nothing is dirty in the L2 when the kernel ends
__device__ __forceinline__ void st16(void *dst, uint4 v, int wt)
{
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    if (wt) {
        const u32x4_t q = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(q));
    } else *reinterpret_cast<uint4 *>(dst) = v;
}
__device__ __forceinline__ void st16f(float4 *dst, float4 v, int wt)
{
    st16(dst, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)), wt);
}

// exact-GELU's erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the result): ~14 VALU
// instructions instead of erff()'s ~45 with branches -- in the fc1 GEMM (6.3 M evaluations) the libm form was 40 % of the
// kernel (in-situ ablation, tools/gemm_ablate.py)
__device__ __forceinline__ float gelu_erf(float v)
{
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
    const float erf_abs = 1.0f - poly * t * e;            // erf(|v| / sqrt 2)
    return 0.5f * (v + fabsf(v) * erf_abs);               // 0.5 v (1 + sign(v) erf(|v|/sqrt 2))
}

// two values at a time: the polynomial, the squares and the final combination are packed-fp32 instructions (v_pk_fma_f32 /
// v_pk_mul_f32: one issue slot for two elements; only rcp and exp2 stay scalar) -- the epilogue of fc1 is 6.3 M evaluations with
// nothing else to overlap them
typedef float gelu_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f2 gelu_erf2(gelu_f2 v)
{
    const gelu_f2 av = gelu_f2{fabsf(v.x), fabsf(v.y)};
    const gelu_f2 x = av * 0.70710678118654752f;
    const gelu_f2 d = x * 0.3275911f + 1.0f;
    const gelu_f2 t = gelu_f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    gelu_f2 poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    const gelu_f2 a = x * x * (-1.4426950408889634f);
    const gelu_f2 e = gelu_f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const gelu_f2 erf_abs = 1.0f - poly * t * e;
    return (v + av * erf_abs) * 0.5f;
}

// The accumulator fragment layout gives a lane 4 consecutive columns per fragment (n = i*16 + g*4 + r for row m = lane&15);
// one v_permlane16_swap per register between the fragments of a pair (2q, 2q+1) turns that into 8 consecutive columns per
// lane, n = q*32 + (g&1)*16 + (g>>1)*8 + e, so the four lanes of a row cover 32 consecutive columns: whole 128-byte lines
// of the fp32 residual stream (two float4 per lane) and 16-byte bf16 stores, instead of 64- / 32-byte pieces per row.
__device__ __forceinline__ void pair_exchange(const f32x4 &a, const f32x4 &b, float (&w)[8])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[r]), __float_as_uint(b[r]), false, false);
        w[r] = __uint_as_float(sw[0]);
        w[4 + r] = __uint_as_float(sw[1]);
    }
}

// residual-stream operands of the gated-accumulate epilogue in the exchanged layout, fetched before the K loop
// (mrow0 / ncol0: first row / column of the WAVE's part of the tile)
template <int MT>
struct ResidualPrefetch {
    f32x4 x[MT][2][2], gate[MT][2][2];
};

template <int MT, int NQ = 2>
__device__ __forceinline__ void residual_prefetch(const GemmP &p, ResidualPrefetch<MT> &pf, int mrow0, int ncol0, int lane)
{
    const int g = lane >> 4;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = min(mrow0 + j * 16 + (lane & 15), p.M - 1);
        const float *gate_row = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.gate_stride : nullptr;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int n = min(ncol0 + q * 32 + (g & 1) * 16 + (g >> 1) * 8 + hh * 4, p.N - 4);
                const float4 xv = *reinterpret_cast<const float4 *>(static_cast<const float *>(p.out) + (size_t)m * p.ldo + n);
                pf.x[j][q][hh] = f32x4{xv.x, xv.y, xv.z, xv.w};
                float4 gv = make_float4(1.f, 1.f, 1.f, 1.f);
                if (gate_row) gv = *reinterpret_cast<const float4 *>(gate_row + n);
                pf.gate[j][q][hh] = f32x4{gv.x, gv.y, gv.z, gv.w};
            }
    }
}

// the consumer's row sums of squares (<= 16 partials per row), requested before the K loop like the residual rows
// (rows of the partial sums are ss_ld(tiles) floats apart: the tile count rounded up to a multiple of 4, the pad entries written as zeros by
//  the producer -- 18 tiles at width 1152 -> 20; NP float4 groups per row: 4 for the widths the four-slot kernels see, <= 1024, 5 in the
//  three-slot ones, <= 1280)
__host__ __device__ __forceinline__ int ss_ld(int tiles) { return (tiles + 3) & ~3; }

template <int MT, int NP = 4>
struct RowSsPrefetch {
    f32x4 part[MT][NP];
};

template <int MT, int NP = 4>
__device__ __forceinline__ void rowss_prefetch(const GemmP &p, RowSsPrefetch<MT, NP> &pf, int mrow0, int lane)
{
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = min(mrow0 + j * 16 + (lane & 15), p.M - 1);
        const float *rp = p.row_ss + (size_t)m * ss_ld(p.row_ss_tiles);
#pragma unroll
        for (int t = 0; t < NP; ++t) {
            pf.part[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (4 * t < p.row_ss_tiles) {
                const float4 v = *reinterpret_cast<const float4 *>(rp + 4 * t);
                pf.part[j][t] = f32x4{v.x, v.y, v.z, v.w};
            }
        }
    }
}

// FN = 16-column fragments per wave: 4 (the wave's 64 columns are one attention head / one emit_ss group) or 2 (32 columns: no
// per-head RMSNorm, no V^T store; the row sums of squares of a 64-column group are returned in `emit_part` for the caller to
// combine across the two waves of the group)
template <int EPI, int MT, int FN, bool PRE, bool RSSPRE, bool PREONLY = false, int NP = 4>
__device__ __forceinline__ void gemm_epilogue(const GemmP &p, f32x4 (&acc)[FN][MT], const ResidualPrefetch<MT> *pre,
                                              const RowSsPrefetch<MT, NP> *rss, int mrow0, int ncol0, int lane, float *emit_part = nullptr,
                                              const f32x4 *bias_pre = nullptr, const f32x4 *qkw_pre = nullptr,
                                              float *qk_mine = nullptr, const float *qk_other = nullptr,
                                              const f32x4 (*emul_pre)[2] = nullptr)
{
    // PREONLY (the ring kernels): every per-column operand -- bias row, emit multipliers -- was requested before the K loop and NO load
    // is compiled into the row loop below.  A load that MAY have been issued makes the compiler wait for vmcnt(0) where the paths
    // join, and on gfx950 that counter also holds the output stores of the previous row fragment: the epilogue then runs one row
    // fragment per memory round trip (measured: every GEMM of the evaluation 0.5 ... 2 us longer).  The dispatcher sends shapes whose
    // waves straddle two batch items of a per-batch operand to the general kernel instead.
    // (never select between the caller's register array and nullptr at run time: that sends the array to scratch)
    static_assert(FN == 4 || FN == 2, "a wave owns 64 or 32 columns");
    const int M = p.M, N = p.N;
    // epilogue: lane holds acc[i][j][r] = C[m = mrow0 + j*16 + (lane&15)][n = ncol0 + i*16 + (lane>>4)*4 + r];
    // with FN = 4 the 64 columns of a wave are one attention head: its 64 values of row m sit in 4 fragments x 4 lane-groups x 4
    // registers, so the per-head RMSNorm is an in-lane sum plus two xor-shuffles
    // (FN = 2: the wave holds one half of the head; the two waves of the 64-column workgroup tile exchange their halves of the sum
    //  of squares through LDS -- qk_mine / qk_other, MT * 16 floats each -- with one workgroup barrier)
    const int nhead = ncol0, g = lane >> 4;
    const float *qkw = nullptr;
    if (EPI == GA_GEMM_EPI_STORE_BF16) {
        if (nhead < p.qk_cols0) qkw = p.qk_w0;
        else if (nhead < p.qk_cols1) qkw = p.qk_w1;
        if (qkw && FN == 2) qkw += nhead & 63;
    }
    const bool to_vt = EPI == GA_GEMM_EPI_STORE_BF16 && p.vt && nhead >= p.vt_col0;  // wave-uniform (vt_col0 % 64 == 0)
    // bias and the folded row scale of row fragment j, applied to v
    auto prepare = [&](int j, f32x4 (&v)[FN]) __attribute__((always_inline)) {
        const int m = mrow0 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < FN; ++i) v[i] = acc[i][j];
        if (EPI == GA_GEMM_EPI_RESIDUAL && p.k_rows && m >= p.k_rows) {   // rows behind the product: bias (and emit) only
#pragma unroll
            for (int i = 0; i < FN; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if ((EPI == GA_GEMM_EPI_STORE_BF16 || EPI == GA_GEMM_EPI_GELU_BF16) && p.row_ss) {  // kernel-uniform: the RMSNorm row scale folded out of the A operand
            float tot = 0.f;
            if (RSSPRE) {
#pragma unroll
                for (int t = 0; t < NP; ++t) tot += (rss->part[j][t][0] + rss->part[j][t][1]) + (rss->part[j][t][2] + rss->part[j][t][3]);
            } else {  // two-workgroups-per-CU configuration: no registers to spare for a prefetch, same summation order
                const float *rp = p.row_ss + (size_t)min(m, M - 1) * ss_ld(p.row_ss_tiles);
                for (int t = 0; t < p.row_ss_tiles; t += 4) {
                    const float4 q4 = *reinterpret_cast<const float4 *>(rp + t);
                    tot += (q4.x + q4.y) + (q4.z + q4.w);
                }
            }
            const float rs = rsqrtf(tot * p.row_ss_inv_dim + p.row_ss_eps);
#pragma unroll
            for (int i = 0; i < FN; ++i) { v[i][0] *= rs; v[i][1] *= rs; v[i][2] *= rs; v[i][3] *= rs; }
        }
        // (the bias after the row scale: with a folded modulated norm it carries shift_b W^T, which is not scaled)
        const float *brow = p.bias;
        if (!PREONLY && !bias_pre && p.bias && p.bias_stride) brow += (size_t)(min(m, M - 1) / p.rows_per_batch) * p.bias_stride;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int n = nhead + i * 16 + g * 4;
            if (PREONLY || bias_pre) {            // requested before the K loop (zero without a bias)
                v[i][0] += bias_pre[i][0]; v[i][1] += bias_pre[i][1]; v[i][2] += bias_pre[i][2]; v[i][3] += bias_pre[i][3];
            } else if (p.bias && n < N) {
                const float4 b = *reinterpret_cast<const float4 *>(brow + n);
                v[i][0] += b.x; v[i][1] += b.y; v[i][2] += b.z; v[i][3] += b.w;
            }
        }
    };
    const bool halves = EPI == GA_GEMM_EPI_STORE_BF16 && FN == 2 && qk_mine != nullptr;   // workgroup-uniform
    if (halves) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            f32x4 v[FN];
            prepare(j, v);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                acc[i][j] = v[i];
                ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            }
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (g == 0) qk_mine[j * 16 + (lane & 15)] = ss;
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = mrow0 + j * 16 + (lane & 15);
        f32x4 v[FN];
        if (halves) {
#pragma unroll
            for (int i = 0; i < FN; ++i) v[i] = acc[i][j];
        } else {
            prepare(j, v);
        }
        if (EPI == GA_GEMM_EPI_STORE_BF16 && qkw && (FN == 4 || halves)) {  // wave-uniform
            float ss;
            if (FN == 4) {
                ss = 0.f;
#pragma unroll
                for (int i = 0; i < FN; ++i) ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
            } else {   // (lower half first: the same sum in both waves)
                const float mine = qk_mine[j * 16 + (lane & 15)], other = qk_other[j * 16 + (lane & 15)];
                ss = (nhead & 32) ? other + mine : mine + other;
            }
            const float rs = rsqrtf(ss * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                float4 w;
                if (qkw_pre) w = make_float4(qkw_pre[i][0], qkw_pre[i][1], qkw_pre[i][2], qkw_pre[i][3]);
                else w = *reinterpret_cast<const float4 *>(qkw + i * 16 + g * 4);
                v[i][0] *= rs * w.x; v[i][1] *= rs * w.y; v[i][2] *= rs * w.z; v[i][3] *= rs * w.w;
            }
        }
        if (EPI == GA_GEMM_EPI_GELU_BF16) {
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const gelu_f2 lo = gelu_erf2(gelu_f2{v[i][0], v[i][1]}), hi = gelu_erf2(gelu_f2{v[i][2], v[i][3]});
                v[i] = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
        }
        if (to_vt) {
            // V projection: write V^T[(b*heads + h)*64 + d][token]; 16 consecutive lanes hold 16 consecutive tokens.  (A 4x4
            // transpose inside lane quads -- three DPP moves, 8-byte stores of 4 tokens -- was measured: 24.4 vs 22.8 us for the
            // qkv GEMM; the 2-byte stores of 16 consecutive tokens coalesce well enough and cost no extra VALU.)
            if (m >= M) continue;
            const int b = m / p.rows_per_batch, tok = m - b * p.rows_per_batch;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const int n = nhead + i * 16 + g * 4;
                if (n >= N) continue;
                uint16_t *dst = p.vt + ((size_t)b * p.heads * 64 + (n - p.vt_col0)) * p.vt_ld + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(size_t)r * p.vt_ld] = f32_to_bf16(v[i][r]);
            }
            continue;
        }
        float emit_acc = 0.f;  // this lane's share of sum x_new^2 over the wave's columns of row m
#pragma unroll
        for (int q = 0; q < FN / 2; ++q) {
            float w[8];
            pair_exchange(v[2 * q], v[2 * q + 1], w);   // every lane takes part, also rows m >= M
            const int n = nhead + q * 32 + (g & 1) * 16 + (g >> 1) * 8;
            if (m >= M || n >= N) continue;
            const bool hi = n + 4 < N;                   // N % 4 == 0: the second half of the 8 may lie past the edge
            if (EPI == GA_GEMM_EPI_STORE_BF16 || EPI == GA_GEMM_EPI_GELU_BF16) {
                uint16_t *dst = static_cast<uint16_t *>(p.out) + (size_t)m * p.ldo + n;
                const uint2 lo = make_uint2(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]));
                const uint2 hi2 = make_uint2(pack_bf16x2(w[4], w[5]), pack_bf16x2(w[6], w[7]));
                if (hi && (p.ldo & 7) == 0) st16(dst, make_uint4(lo.x, lo.y, hi2.x, hi2.y), p.wt);
                else {
                    *reinterpret_cast<uint2 *>(dst) = lo;
                    if (hi) *reinterpret_cast<uint2 *>(dst + 4) = hi2;
                }
            } else {
                float4 *dst = reinterpret_cast<float4 *>(static_cast<float *>(p.out) + (size_t)m * p.ldo + n);
                if (EPI == GA_GEMM_EPI_RESIDUAL) {
                    f32x4 x0, x1, g0 = f32x4{1.f, 1.f, 1.f, 1.f}, g1 = g0;
                    if (PRE) {
                        x0 = pre->x[j][q][0]; x1 = pre->x[j][q][1]; g0 = pre->gate[j][q][0]; g1 = pre->gate[j][q][1];
                    } else {
                        const float4 a0 = dst[0], a1 = hi ? dst[1] : make_float4(0.f, 0.f, 0.f, 0.f);
                        x0 = f32x4{a0.x, a0.y, a0.z, a0.w}; x1 = f32x4{a1.x, a1.y, a1.z, a1.w};
                        if (p.gate) {
                            const float *gate_row = p.gate + (size_t)(m / p.rows_per_batch) * p.gate_stride + n;
                            const float4 b0 = *reinterpret_cast<const float4 *>(gate_row);
                            const float4 b1 = hi ? *reinterpret_cast<const float4 *>(gate_row + 4) : b0;
                            g0 = f32x4{b0.x, b0.y, b0.z, b0.w}; g1 = f32x4{b1.x, b1.y, b1.z, b1.w};
                        }
                    }
                    float xn[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xn[e] = x0[e] + g0[e] * w[e];
                        xn[4 + e] = hi ? x1[e] + g1[e] * w[4 + e] : 0.f;
                    }
                    st16f(dst, make_float4(xn[0], xn[1], xn[2], xn[3]), p.wt);
                    if (hi) st16f(dst + 1, make_float4(xn[4], xn[5], xn[6], xn[7]), p.wt);
                    if (p.emit_x) {  // kernel-uniform; N % 64 == 0, so `hi` holds
#pragma unroll
                        for (int e = 0; e < 8; ++e) emit_acc += xn[e] * xn[e];
                        if (p.emit_w && (PREONLY || emul_pre)) {   // (requested before the K loop: one batch item per wave)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                xn[e] *= emul_pre[q][0][e] * (1.f + emul_pre[FN / 2 + q][0][e]);
                                xn[4 + e] *= emul_pre[q][1][e] * (1.f + emul_pre[FN / 2 + q][1][e]);
                            }
                        } else if (!PREONLY && p.emit_w) {   // kernel-uniform: the modulated pre-norm's w (1 + scale_b), applied before the bf16 rounding
                            const float *sc = p.emit_scale + (size_t)(m / p.rows_per_batch) * p.emit_scale_stride + n;
                            const float4 w0 = *reinterpret_cast<const float4 *>(p.emit_w + n), w1 = *reinterpret_cast<const float4 *>(p.emit_w + n + 4);
                            const float4 s0 = *reinterpret_cast<const float4 *>(sc), s1 = *reinterpret_cast<const float4 *>(sc + 4);
                            xn[0] *= w0.x * (1.f + s0.x); xn[1] *= w0.y * (1.f + s0.y); xn[2] *= w0.z * (1.f + s0.z); xn[3] *= w0.w * (1.f + s0.w);
                            xn[4] *= w1.x * (1.f + s1.x); xn[5] *= w1.y * (1.f + s1.y); xn[6] *= w1.z * (1.f + s1.z); xn[7] *= w1.w * (1.f + s1.w);
                        }
                        st16(p.emit_x + (size_t)m * p.emit_ld + n,
                             make_uint4(pack_bf16x2(xn[0], xn[1]), pack_bf16x2(xn[2], xn[3]), pack_bf16x2(xn[4], xn[5]),
                                        pack_bf16x2(xn[6], xn[7])), p.wt);
                    }
                } else {
                    dst[0] = make_float4(w[0], w[1], w[2], w[3]);
                    if (hi) dst[1] = make_float4(w[4], w[5], w[6], w[7]);
                }
            }
        }
        if (EPI == GA_GEMM_EPI_RESIDUAL && p.emit_ss) {  // kernel-uniform: the four lanes of a row add up in a fixed order
            emit_acc += __shfl_xor(emit_acc, 16, 64);
            emit_acc += __shfl_xor(emit_acc, 32, 64);
            if (FN == 4) {
                if (g == 0 && m < M && nhead < N) {
                    float *er = p.emit_ss + (size_t)m * ss_ld(N >> 6);
                    er[nhead >> 6] = emit_acc;
                    if ((nhead >> 6) + 1 == (N >> 6))            // the last group's writer zeroes the pad entries of the row
                        for (int z = N >> 6; z < ss_ld(N >> 6); ++z) er[z] = 0.f;
                }
            } else if (g == 0) emit_part[j * 16 + (lane & 15)] = emit_acc;   // half a group: the caller adds the two waves up
        }
    }
}

// MT = activation-row fragments per wave: 4 -> 128-row tiles, 2 -> 64-row tiles (twice the workgroups for the N = 1024
// GEMMs, whose 128x128 grid fills only 96 of the 256 CUs)
template <int EPI, int NST, int MT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];  // [NSTAGE][W | A][row][slot] = 4 x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    constexpr int BMT = 2 * MT * 16;                       // activation rows per workgroup tile
    constexpr int SLOT = (BN + BMT) * BK;                  // elements of one ring slot (W tile + A tile)
    constexpr int DMA_PER_TILE = 4 + MT;                   // DMA instructions per wave per K-tile
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BMT;
    const int M = p.M, N = p.N, K = p.K;

    // DMA assignment: instruction i of wave w fills rows (w*4+i)*8 .. +7; lane -> row + (lane>>3), slot lane&7,
    // which receives global chunk (lane&7) ^ (row&7)
    const uint16_t *srcW[4], *srcA[MT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        srcW[i] = p.W + w_offset(p, min(n0 + row, N - 1), (lane & 7) ^ (row & 7));
    }
    const int wks = w_kstep(p);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = (wave * MT + i) * 8 + (lane >> 3);
        srcA[i] = p.A + (size_t)min(m0 + row, ((EPI == GA_GEMM_EPI_RESIDUAL && p.k_rows) ? p.k_rows : M) - 1) * p.lda + ((lane & 7) ^ (row & 7)) * 8;
    }
    f32x4 acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the residual rows and gates of the gated-accumulate epilogue are requested before the K loop (they are older than every
    // DMA, so the counted vmcnt waits below still mean what they say); only with one workgroup per CU, where the 64-128
    // extra VGPRs are free
    constexpr bool PRE = EPI == GA_GEMM_EPI_RESIDUAL && NST >= 4;
    ResidualPrefetch<PRE ? MT : 1> pre;
    const int mrow0 = m0 + wm * (MT * 16), ncol0 = n0 + wn * 64;
    if (PRE) residual_prefetch<MT>(p, reinterpret_cast<ResidualPrefetch<MT> &>(pre), mrow0, ncol0, lane);
    constexpr bool RSS = EPI == GA_GEMM_EPI_STORE_BF16 && NST >= 4;   // not in the 2-workgroups-per-CU configuration (VGPR budget 256)
    RowSsPrefetch<RSS ? MT : 1> rss;
    if (RSS && p.row_ss) rowss_prefetch<MT>(p, reinterpret_cast<RowSsPrefetch<MT> &>(rss), mrow0, lane);

    const int frow = lane & 15, g = lane >> 4;
    const int nk = K / BK;

    // Buffer indices are compile-time constants in every use below (the K loop is unrolled by two): with a run-time
    // buffer index hipcc cannot separate the DMA destination from the fragment reads and drains the DMA (vmcnt(0))
    // in front of every ds_read, which serialises load and math.
#define GA_STAGE(BUF, KT)                                                                             \
    do {                                                                                              \
        uint16_t *bw_ = smem + (BUF) * SLOT, *ba_ = bw_ + TILE_ELEMS;                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
            glds16(srcW[i] + (size_t)(KT) * wks, bw_ + (wave * 4 + i) * 8 * BK);                       \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                \
            glds16(srcA[i] + (size_t)(KT) * BK, ba_ + (wave * MT + i) * 8 * BK);                      \
    } while (0)
#define GA_COMPUTE(BUF)                                                                               \
    do {                                                                                              \
        /* all fragment reads of the K-tile go out in one batch (pinned), then the MFMAs */            \
        const uint16_t *bw_ = smem + (BUF) * SLOT, *ba_ = bw_ + TILE_ELEMS;                            \
        bf16x8 fw[2][4], fa[2][MT];                                                                   \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                            \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                const int rw = wn * 64 + i * 16 + frow;                                               \
                if (GA_GEMM_ABLATE & 16) fw[kk][i] = bf16x8{(short)rw, 1, 2, 3, 4, 5, 6, 7};              \
                else fw[kk][i] = *reinterpret_cast<const bf16x8 *>(bw_ + rw * BK + (((kk * 4 + g) ^ (rw & 7)) * 8)); \
            }                                                                                         \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                          \
                const int ra = wm * (MT * 16) + i * 16 + frow;                                        \
                if (GA_GEMM_ABLATE & 16) fa[kk][i] = bf16x8{(short)ra, 1, 2, 3, 4, 5, 6, 7};              \
                else fa[kk][i] = *reinterpret_cast<const bf16x8 *>(ba_ + ra * BK + (((kk * 4 + g) ^ (ra & 7)) * 8)); \
            }                                                                                         \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
                _Pragma("unroll") for (int j = 0; j < MT; ++j)                                        \
                    if (!(GA_GEMM_ABLATE & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[kk][i], fa[kk][j], acc[i][j], 0, 0, 0); \
                    else acc[i][j][0] += __builtin_bit_cast(float, (int)fw[kk][i][0] ^ (int)fa[kk][j][0]); \
    } while (0)

    // ring of NST slots: tiles kt+1 .. kt+NST-1 are in flight while tile kt is multiplied (NST-1 tiles of look-ahead).
    // NST = 4 (128 KiB, one workgroup per CU) for the small grids, NST = 2 (64 KiB, two workgroups per CU, which hide each
    // other's latency) when the grid has more than one workgroup per CU -- chosen by the host from the grid size.
    // What bounds the K loop at the DiT shapes is LDS traffic, not latency or the matrix pipe (in-situ ablation,
    // tools/gemm_ablate.py, N = 1024, K = 4096: 30.3 us; without MFMAs 29.2; without the fragment reads 22.9; without the
    // DMA 22.1; without all three 12.7): a 128 x 64 tile reads 48 KiB of fragments and takes 24 KiB of DMA writes per
    // 272 cycles of MFMA work.  Reading the next tile's fragments under the current MFMAs (register double buffer) and a
    // 6-slot ring were both measured: no gain.
#define GA_WAIT_TILES_IN_FLIGHT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((N) * DMA_PER_TILE) : "memory")
#define GA_PHASE(BUF, KT)                                                                              \
    do {                                                                                               \
        const int rem_ = nk - 1 - (KT); /* tiles after KT; at most NST-2 of them are issued so far */    \
        if (NST >= 4 && rem_ >= 2) GA_WAIT_TILES_IN_FLIGHT(2);                                          \
        else if (NST >= 3 && rem_ >= 1) GA_WAIT_TILES_IN_FLIGHT(1);                                     \
        else GA_WAIT_TILES_IN_FLIGHT(0);                                                                \
        if (!(GA_GEMM_ABLATE & 4)) __builtin_amdgcn_s_barrier(); /* everyone's part of tile KT landed; everyone left tile KT-1 */  \
        if (!(GA_GEMM_ABLATE & 2) && (KT) + NST - 1 < nk) GA_STAGE(((BUF) + NST - 1) % NST, (KT) + NST - 1); \
        GA_COMPUTE(BUF);                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* my fragment reads of this buffer are done */ \
    } while (0)

    GA_STAGE(0, 0);
    if (GA_GEMM_ABLATE == 32) {  // launch + prologue + epilogue only
        gemm_epilogue<EPI, MT, 4, PRE, RSS>(p, acc, reinterpret_cast<const ResidualPrefetch<MT> *>(&pre),
                                reinterpret_cast<const RowSsPrefetch<MT> *>(&rss), mrow0, ncol0, lane);
        return;
    }
    if (NST > 2 && nk > 1) GA_STAGE(1 % NST, 1);
    if (NST > 3 && nk > 2) GA_STAGE(2 % NST, 2);
    for (int kt = 0; kt < nk; kt += NST) {
        GA_PHASE(0, kt);
        if (kt + 1 < nk) GA_PHASE(1 % NST, kt + 1);
        if (NST > 2 && kt + 2 < nk) GA_PHASE(2 % NST, kt + 2);
        if (NST > 3 && kt + 3 < nk) GA_PHASE(3 % NST, kt + 3);
    }
#undef GA_PHASE
#undef GA_WAIT_TILES_IN_FLIGHT
#undef GA_STAGE
#undef GA_COMPUTE

    if (GA_GEMM_ABLATE & 8) {  // one store per lane instead of the epilogue
        float t = 0.f;
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < MT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.f) static_cast<float *>(p.out)[tid] = t;
        return;
    }
    gemm_epilogue<EPI, MT, 4, PRE, RSS>(p, acc, reinterpret_cast<const ResidualPrefetch<MT> *>(&pre),
                                reinterpret_cast<const RowSsPrefetch<MT> *>(&rss), mrow0, ncol0, lane);
}

// ---- round 3: the ring kernel -------------------------------------------------------------------------------------------
// What the test bed (tools/gemm_lab.hip: every variant checked on the full output, timed with cold weights) showed at the
// DiT shapes, M = 1536 rows:
//   * all tilings of the wide GEMMs (fc1, qkv) sit between 17 and 25 us: the aggregate L2 -> CU operand stream saturates at
//     about 16 TB/s (21-26 B/clk/CU with every CU loading), so time ~ total operand bytes through the L1s, i.e. the tile with
//     the fewest bytes per flop that still puts one workgroup on every CU wins: 192 x 128 (256 / 192 workgroups);
//   * one wave per SIMD leaves the matrix pipe idle whenever that wave waits (barrier, DMA issue, fragment latency): EIGHT waves
//     (4 x 2, wave tile 48 x 64) on the 192 x 128 tile beat four waves of 96 x 64 by 10 % although they read more LDS bytes;
//   * hipcc on gfx950 only ever emits `s_waitcnt lgkmcnt(0)`, so a software pipeline written in HIP waits for the fragment reads
//     it has just issued for the NEXT k-step: the reads are inline asm here and the waits are counted by hand (lgkmcnt(7));
//   * the K loop has no run-time branch: the ring slot, the wait count and "is there a tile left to request" are compile-time
//     constants of a peeled tail, so the steady state is one basic block (every block boundary costs an lgkmcnt(0) drain);
//   * the N = 1024 GEMMs (proj, fc2, cross-attention q / out) were latency-bound with a 2-slot ring: 96 x 64 and 64 x 64 tiles
//     with FOUR slots run proj in 6.7 us (was 11.6), fc2 in 24 us (was 33), the M = 768 projections in 5.3 us (was 9).
//   * measured without gain: 32x32x16 MFMAs, 256 x 128 / 256 x 256 / 192 x 192 tiles, 8-slot rings, padded row pitches
//     (no channel camping), weights stored as contiguous 1 KiB DMA blocks (3-6 % on cold weights only), output staged through
//     LDS into whole 256-byte rows (the store tail of a GEMM is the L2 write-back at the kernel boundary, not coalescing).
// WM x WN waves, each FM x FN fragments of 16 x 16; ring of NST slots of (BN + BM) x 64 bf16; PIPE 0: all fragment reads of a
// K-tile in one batch, then its MFMAs; PIPE 2: k-step software pipeline with asm reads.  Requires K % (64 NST) == 0, K >= 128 NST.
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

#define GA_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// SK > 0: split-K over SK workgroups per tile (blockIdx.z).  Every workgroup writes its fp32 accumulator fragments lane-contiguously
// (1 KiB per store instruction) with agent-scope stores, waits for them, and counts itself on the tile's word; the workgroup that
// finds SK - 1 there reads the partials back with agent-scope loads and adds them IN SPLIT ORDER ((p0 + p1) + p2) + p3 -- its own
// from registers, the same bits it stored -- so the result does not depend on who arrives last.  No fence (an agent-scope fence
// writes back / invalidates the whole L2 of the XCD, see surfel_bin.hip: merge_runs); nobody waits for anybody: no residency assumption.
// R: K-tiles beyond a multiple of NST (round 6: K = 1152 = 18 tiles runs the FOUR-slot ring with R = 2 -- the peeled tail is NST + R tiles
// long; on the three-slot ring the same GEMMs ran 1.5 x longer, two tiles in flight per DMA latency instead of three)
template <int EPI, int WM, int WN, int FM, int FN, int NST, int PIPE, int SK = 0, int R = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_ring_kernel(GemmP p)
{
    static_assert(R >= 0 && R < NST && (R == 0 || SK == 0), "remainder tiles");
    constexpr int TAIL = NST + R;
    constexpr int NW = WM * WN, BM = WM * FM * 16, BNT = WN * FN * 16;
    constexpr int KS = 2, CPK = 4;            // k-steps of 32 per K-tile, 16-byte chunks per k-step
    constexpr int ROWS = BNT + BM;
    static_assert(ROWS % (8 * NW) == 0, "DMA rows must divide among the waves");
    constexpr int DPT = ROWS / 8 / NW;        // DMA instructions per wave per K-tile
    constexpr int SLOT = ROWS * BK;           // elements per ring slot
    static_assert((NST - 2) * DPT <= 63, "vmcnt range");
    static_assert(NST >= 3 || PIPE == 0, "the k-step pipeline hands over to a tile requested one tile earlier");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WM, wm = wave % WM;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xmap) {   // kernel-uniform (host: gridDim.x % 2 == 0, gridDim.y % 4 == 0)
        const int lin = by * (int)gridDim.x + bx, c = lin & 7, k = lin >> 3, bw = (int)gridDim.x >> 1, bh = (int)gridDim.y >> 2;
        bx = (c & 1) * bw + k % bw;
        by = (c >> 1) * bh + k / bw;
    }
    const int n0 = bx * BNT, m0 = by * BM;
    const int M = p.M, N = p.N, K = p.K;
    const int nk = K / BK / (SK > 0 ? SK : 1);                 // K-tiles of THIS workgroup
    const int kt0 = SK > 0 ? (int)blockIdx.z * nk : 0;         // its first one

    // DMA sources: instruction i of this wave fills rows q*8 .. q*8+7 of the slot image [W rows | A rows], q = i*NW + wave;
    // slot (row, s) of the 128-byte row receives global chunk s ^ (row & 7)
    // (round 6: the SADDR form the attention's key walk gained 1 - 2 % from -- wave-uniform base + a 32-bit lane offset computed once -- was
    //  built here too, bases through readfirstlane and the zero-extension pinned next to its use or LLVM folds it back into one 64-bit pointer
    //  per lane: all DMA issues in SADDR form, DiT-L 2.776 vs 2.778 ms per evaluation same-box: nothing.  One v_lshl_add_u64 per DMA
    //  instruction was all there was to save, and these loops are not issue-bound.  Not kept.)
    const uint16_t *src[DPT];
    static_assert(BNT % (8 * NW) == 0, "instruction i of every wave is on the same side of the W | A boundary");
    const int arows = (EPI == GA_GEMM_EPI_RESIDUAL && p.k_rows) ? p.k_rows : M;   // rows of A that exist / take part
    const bool product = m0 < arows;                                              // workgroup-uniform: a tile behind them skips the K loop
    const int wks = w_kstep(p);
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
        const int row = (i * NW + wave) * 8 + (lane >> 3);
        if (i * NW * 8 < BNT) src[i] = p.W + w_offset(p, min(n0 + row, N - 1), (lane & 7) ^ (row & 7)) + (size_t)kt0 * wks;
        else src[i] = p.A + (size_t)min(m0 + row - BNT, arows - 1) * p.lda + ((lane & 7) ^ ((row - BNT) & 7)) * 8 + (size_t)kt0 * BK;
    }
    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, lg = lane >> 4;
    // element offset of this lane's 16-byte piece inside a row-major [rows][64] tile at k-step 0; k-step ks flips bit 2 of the chunk
    const int lane_off = lrow * BK + ((lg ^ (lrow & 7)) * 8);
    const int mrow0 = m0 + wm * FM * 16, ncol0 = n0 + wn * FN * 16;

    auto stage = [&](auto bufc, int kt) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        uint16_t *base = smem + BUF * SLOT;
#pragma unroll
        for (int i = 0; i < DPT; ++i) glds16(src[i] + (size_t)kt * (i * NW * 8 < BNT ? wks : BK), base + (i * NW + wave) * 8 * BK);
    };
    auto mfmas = [&](const bf16x8(&fw)[FN], const bf16x8(&fa)[FM]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
    };
    // Epilogue operands requested before the K loop (they are older than every DMA, so the counted vmcnt waits still mean what
    // they say): bias, per-head norm weights, the consumer's row sums, and -- where the registers are there (the 4-wave tiles) --
    // the residual rows and gates.  Without this the epilogue starts with a dependent L2 / HBM round trip per operand.
    f32x4 bias_pre[FN], qkw_pre[FN];
    const float *bias_row = p.bias;
    // (a per-batch bias: all rows of a wave lie in one batch item -- rows_per_batch % (FM * 16) == 0, checked by the dispatcher)
    if (p.bias && p.bias_stride) bias_row += (size_t)(min(mrow0, M - 1) / p.rows_per_batch) * p.bias_stride;
    // (no branch around these loads: an absent operand is read from a zero line instead.  With branches the register allocator
    //  merged the two paths through copies of the loaded registers -- s_waitcnt vmcnt(0) in front of the first DMA request.)
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        qkw_pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float4 b = *reinterpret_cast<const float4 *>(p.bias ? bias_row + min(ncol0 + i * 16 + lg * 4, N - 4) : g_zero_line);
        bias_pre[i] = f32x4{b.x, b.y, b.z, b.w};
    }
    if (EPI == GA_GEMM_EPI_STORE_BF16) {
        const float *qkw = ncol0 < p.qk_cols0 ? p.qk_w0 : (ncol0 < p.qk_cols1 ? p.qk_w1 : nullptr);
        if (qkw) {
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                const float4 w = *reinterpret_cast<const float4 *>(qkw + (ncol0 & 63) + i * 16 + lg * 4);
                qkw_pre[i] = f32x4{w.x, w.y, w.z, w.w};
            }
        }
    }
    // the emit multipliers w (1 + scale_b) of a folded modulated pre-norm, in the exchanged 8-column layout of the epilogue
    constexpr int EQ = EPI == GA_GEMM_EPI_RESIDUAL ? FN / 2 : 1;
    f32x4 emul[2 * EQ][2];     // [q]: norm weight, [EQ + q]: scale -- combined in the epilogue (no wait on these loads before the K loop)
#pragma unroll
    for (int q = 0; q < 2 * EQ; ++q) emul[q][0] = emul[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (EPI == GA_GEMM_EPI_RESIDUAL) {
        const bool em = p.emit_w != nullptr;   // kernel-uniform
        const float *sc = em ? p.emit_scale + (size_t)(min(mrow0, M - 1) / max(p.rows_per_batch, 1)) * p.emit_scale_stride : nullptr;
#pragma unroll
        for (int q = 0; q < EQ; ++q) {
            const int n = min(ncol0 + q * 32 + (lg & 1) * 16 + (lg >> 1) * 8, N - 8);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const float4 w4 = *reinterpret_cast<const float4 *>(em ? p.emit_w + n + 4 * hh : g_zero_line);
                const float4 s4 = *reinterpret_cast<const float4 *>(em ? sc + n + 4 * hh : g_zero_line);
                emul[q][hh] = f32x4{w4.x, w4.y, w4.z, w4.w};
                emul[EQ + q][hh] = f32x4{s4.x, s4.y, s4.z, s4.w};
            }
        }
    }
    constexpr bool PRE = EPI == GA_GEMM_EPI_RESIDUAL && NW == 4 && SK == 0;      // 8-wave tile: 256-register budget, no room; split-K: only
                                                                                 // one of SK workgroups runs the epilogue
    ResidualPrefetch<PRE ? FM : 1> pre;
    if (PRE) residual_prefetch<FM, FN / 2>(p, reinterpret_cast<ResidualPrefetch<FM> &>(pre), mrow0, ncol0, lane);
    constexpr bool RSS = EPI == GA_GEMM_EPI_STORE_BF16 || EPI == GA_GEMM_EPI_GELU_BF16;
    constexpr int NP = (NST == 3 || R != 0) ? 5 : 4;      // (the instances that serve the widths that are no multiple of 256: up to 20 partial sums per row)
    RowSsPrefetch<RSS ? FM : 1, NP> rss;
    if (RSS && p.row_ss) rowss_prefetch<FM, NP>(p, reinterpret_cast<RowSsPrefetch<FM, NP> &>(rss), mrow0, lane);

    // nk = n_main * NST + TAIL: the last TAIL = NST + R tiles are peeled (compile-time slot, wait count, request-or-not)
    const int n_main = nk / NST - 1;
    if (product) {
    static_for<0, NST - 1>([&](auto bc) __attribute__((always_inline)) { stage(bc, decltype(bc)::value); });

    if constexpr (PIPE == 0) {
        auto tile = [&](auto bc, int kt, auto stagec, auto flyc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            GA_WAIT_VM(decltype(flyc)::value * DPT);
            __builtin_amdgcn_s_barrier();   // everyone's part of tile kt landed; everyone left tile kt-1
            if constexpr (decltype(stagec)::value) stage(std::integral_constant<int, (b + NST - 1) % NST>{}, kt + NST - 1);
            const uint16_t *bw = smem + b * SLOT, *ba = bw + BNT * BK;
            bf16x8 fw[KS][FN], fa[KS][FM];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < FN; ++i)
                    fw[ks][i] = *reinterpret_cast<const bf16x8 *>(bw + (wn * FN + i) * 16 * BK + (lane_off ^ (ks * CPK * 8)));
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    fa[ks][j] = *reinterpret_cast<const bf16x8 *>(ba + (wm * FM + j) * 16 * BK + (lane_off ^ (ks * CPK * 8)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mfmas(fw[ks], fa[ks]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        int kt = 0;
        for (int it = 0; it < n_main; ++it) {
            static_for<0, NST>([&](auto bc) __attribute__((always_inline)) {
                tile(bc, kt + decltype(bc)::value, std::true_type{}, std::integral_constant<int, NST - 2>{});
            });
            kt += NST;
        }
        static_for<0, TAIL>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr bool st = i + NST - 1 < TAIL;            // tile kt+i+NST-1 exists
            constexpr int fly = st ? NST - 2 : TAIL - 1 - i;   // issued tiles younger than this one
            tile(std::integral_constant<int, i % NST>{}, kt + i, std::integral_constant<bool, st>{}, std::integral_constant<int, fly>{});
        });
    } else {
        bf16x8 fw[2][FN], fa[2][FM];
        const uint32_t lds0 = (uint32_t)(size_t)(const __attribute__((address_space(3))) uint16_t *)smem;
        uint32_t aw[KS][(NST + 1) / 2], aa[KS][(NST + 1) / 2];   // byte addresses: [k-step][slot pair] (ds_read offsets are 16 bit)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int h = 0; h < (NST + 1) / 2; ++h) {
                aw[ks][h] = lds0 + 2 * (h * 2 * SLOT + wn * FN * 16 * BK + (lane_off ^ (ks * CPK * 8)));
                aa[ks][h] = lds0 + 2 * (h * 2 * SLOT + BNT * BK + wm * FM * 16 * BK + (lane_off ^ (ks * CPK * 8)));
            }
        auto read2 = [&](auto bufc, auto ksc, bf16x8(&w)[FN], bf16x8(&a)[FM]) __attribute__((always_inline)) {
            constexpr int BUF = decltype(bufc)::value, ks = decltype(ksc)::value;
            static_assert(2 * ((BUF & 1) * SLOT + ((FN > FM ? FN : FM) - 1) * 16 * BK) < 65536, "ds_read offset field");
            const uint32_t adw = aw[ks][BUF >> 1], ada = aa[ks][BUF >> 1];
#pragma unroll
            for (int i = 0; i < FN; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[i]) : "v"(adw), "n"(2 * ((BUF & 1) * SLOT + i * 16 * BK)));
#pragma unroll
            for (int j = 0; j < FM; ++j)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j]) : "v"(ada), "n"(2 * ((BUF & 1) * SLOT + j * 16 * BK)));
        };
        // all LDS reads except the NEWEST `newer` have returned; the fragments pass through the asm so that their users cannot be
        // scheduled above the wait
        auto wait2 = [&](auto newerc, bf16x8(&w)[FN], bf16x8(&a)[FM]) __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w[0]) : "n"(decltype(newerc)::value));
#pragma unroll
            for (int i = 1; i < FN; ++i) asm volatile("" : "+v"(w[i]));
#pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(a[j]));
        };
        GA_WAIT_VM((NST - 2) * DPT);
        __builtin_amdgcn_s_barrier();
        read2(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fw[0], fa[0]);
        // HO: 0 last tile, 1 hand over to the next tile without a request, 2 hand over and request tile kt+NST-1
        auto tile = [&](auto bc, int kt, auto hoc, auto flyc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value, HO = decltype(hoc)::value;
            static_for<0, KS>([&](auto ksc) __attribute__((always_inline)) {
                constexpr int ks = decltype(ksc)::value;
                constexpr bool more = ks + 1 < KS || HO > 0;
                if constexpr (ks + 1 < KS) {
                    read2(bc, std::integral_constant<int, ks + 1>{}, fw[(ks + 1) & 1], fa[(ks + 1) & 1]);
                } else if constexpr (HO > 0) {
                    // tile kt+1 must have landed for every wave; everyone has left tile kt-1, whose slot takes tile kt+NST-1
                    GA_WAIT_VM(decltype(flyc)::value * DPT);
                    __builtin_amdgcn_s_barrier();
                    if constexpr (HO == 2) stage(std::integral_constant<int, (b + NST - 1) % NST>{}, kt + NST - 1);
                    read2(std::integral_constant<int, (b + 1) % NST>{}, std::integral_constant<int, 0>{}, fw[0], fa[0]);
                }
                wait2(std::integral_constant<int, more ? FN + FM : 0>{}, fw[ks & 1], fa[ks & 1]);
                mfmas(fw[ks & 1], fa[ks & 1]);
            });
        };
        int kt = 0;
        for (int it = 0; it < n_main; ++it) {
            static_for<0, NST>([&](auto bc) __attribute__((always_inline)) {
                tile(bc, kt + decltype(bc)::value, std::integral_constant<int, 2>{}, std::integral_constant<int, NST - 3>{});
            });
            kt += NST;
        }
        static_for<0, TAIL>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int HO = i == TAIL - 1 ? 0 : (i + NST - 1 < TAIL ? 2 : 1);
            constexpr int fly = HO == 2 ? NST - 3 : (TAIL - i - 2 > 0 ? (TAIL - i - 2 < NST - 3 ? TAIL - i - 2 : NST - 3) : 0);   // issued tiles younger than kt+i+1
            tile(std::integral_constant<int, i % NST>{}, kt + i, std::integral_constant<int, HO>{}, std::integral_constant<int, fly>{});
        });
    }
    }   // product

    if constexpr (SK > 0) {
        const int split = __builtin_amdgcn_readfirstlane((int)blockIdx.z);
        const unsigned tile = (unsigned)by * gridDim.x + (unsigned)bx;
        constexpr size_t FRAG = 64 * 16, WAVE_BYTES = (size_t)FN * FM * FRAG, WG_BYTES = (size_t)NW * WAVE_BYTES;
        char *tile_base = reinterpret_cast<char *>(p.sk_part) + (size_t)tile * SK * WG_BYTES + (size_t)wave * WAVE_BYTES + (size_t)lane * 16;
        // 8-byte agent-scope atomics (global_store / global_load_dwordx2 sc1): COMPILER-VISIBLE memory operations -- the first version
        // used 16-byte inline-asm loads with hand-counted vmcnt and was not launch-to-launch reproducible on the 252-register
        // instantiation (the register allocator may copy an asm output before the wait the compiler knows nothing about)
        auto put = [&](char *dst, const f32x4 &v) __attribute__((always_inline)) {
            unsigned long long *d = reinterpret_cast<unsigned long long *>(dst);
            __hip_atomic_store(d, ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d + 1, ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto get = [&](const char *src_) __attribute__((always_inline)) {
            const unsigned long long *q = reinterpret_cast<const unsigned long long *>(src_);
            const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return f32x4{__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi), __uint_as_float((uint32_t)(hi >> 32))};
        };
        {
            char *mine = tile_base + (size_t)split * WG_BYTES;
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) put(mine + (size_t)(i * FM + j) * FRAG, acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my part of the partial tile has been written ...
        __syncthreads();                                      // ... and everybody's; the ring is idle: its first word carries the rank
        unsigned *rank_word = reinterpret_cast<unsigned *>(smem);
        if (tid == 0) *rank_word = __hip_atomic_fetch_add(p.sk_count + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned rank = *rank_word;
        if (rank != (unsigned)(SK - 1)) return;               // workgroup-uniform
        __syncthreads();                                      // (the epilogues of the 32-column waves reuse the ring's first words)
        if (tid == 0) __hip_atomic_store(p.sk_count + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left clean for the next launch
        // last arriver: the partials in split order, its own from registers (the very bits it stored)
        static_for<0, SK>([&](auto myc) __attribute__((always_inline)) {
            constexpr int MY = decltype(myc)::value;
            if (split != MY) return;
            f32x4 sum[FN][FM];
            static_for<0, SK>([&](auto sc) __attribute__((always_inline)) {
                constexpr int S_ = decltype(sc)::value;
                const char *src_ = tile_base + (size_t)S_ * WG_BYTES;
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) {
                        const f32x4 t = S_ == MY ? acc[i][j] : get(src_ + (size_t)(i * FM + j) * FRAG);
                        if constexpr (S_ == 0) sum[i][j] = t;
                        else sum[i][j] += t;
                    }
            });
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = sum[i][j];
        });
    }

    if constexpr (FN == 4) {
        gemm_epilogue<EPI, FM, 4, PRE, RSS, true, NP>(p, acc, reinterpret_cast<const ResidualPrefetch<FM> *>(&pre),
                                                      reinterpret_cast<const RowSsPrefetch<FM, NP> *>(&rss), mrow0, ncol0, lane, nullptr, bias_pre, qkw_pre, nullptr, nullptr, emul);
    } else {
        // 32-column waves: the two waves of a 64-column group add their row sums of squares through LDS (the ring is idle now)
        static_assert(WN == 2, "a 64-column tile is two 32-column waves");
        float *part = reinterpret_cast<float *>(smem) + (wn * WM + wm) * FM * 16;
        const bool emit = EPI == GA_GEMM_EPI_RESIDUAL && p.emit_ss;   // kernel-uniform
        // the workgroup's 64 columns are a q or k head with a per-head RMSNorm: its two waves exchange their halves of the row sums
        const bool qk2 = EPI == GA_GEMM_EPI_STORE_BF16 && n0 < p.qk_cols1;   // workgroup-uniform
        if (emit || qk2) __syncthreads();
        gemm_epilogue<EPI, FM, 2, PRE, RSS, true, NP>(p, acc, reinterpret_cast<const ResidualPrefetch<FM> *>(&pre),
                                            reinterpret_cast<const RowSsPrefetch<FM, NP> *>(&rss), mrow0, ncol0, lane, part, bias_pre, qkw_pre,
                                            qk2 ? part : nullptr,
                                            reinterpret_cast<const float *>(smem) + ((wn ^ 1) * WM + wm) * FM * 16, emul);
        if (emit) {
            __syncthreads();
            const float *all = reinterpret_cast<const float *>(smem);
            if (tid < BM && m0 + tid < M) {
                float *er = p.emit_ss + (size_t)(m0 + tid) * ss_ld(N >> 6);
                er[n0 >> 6] = all[tid] + all[BM + tid];
                if ((n0 >> 6) + 1 == (N >> 6))
                    for (int z = N >> 6; z < ss_ld(N >> 6); ++z) er[z] = 0.f;
            }
        }
    }
}

// A register-FIFO variant of this kernel (global_load_dwordx4 into D = 4 / 8 K-tiles of VGPRs, ds_write into a double
// buffer) was built and measured in round 1 to test whether more bytes in flight would lift the small-M GEMMs: it does
// not (N = 4096: 46 us vs 29 us; deeper FIFOs slower still).  PMC (tools/pmc_gemm.sh): L1->L2 read latency averages
// 320 cycles and the L1 streams ~25 B/clk/CU, far below its 64 B/clk port -- the miss path of the L1, not the prefetch
// depth, caps these tiles, so only more reuse per byte (larger workgroup tiles on grids that still fill the chip) helps.

}  // namespace gadit

namespace gadit {
static std::atomic<int> &splitk_mode()
{
    static std::atomic<int> mode{[] { const char *e = getenv("GA_GEMM_SPLITK"); return e ? atoi(e) : -1; }()};
    return mode;
}
}  // namespace gadit

extern "C" int ga_gemm_splitk_mode(int mode)
{
    return gadit::splitk_mode().exchange(mode < -1 || mode > 6 ? -1 : mode);
}

extern "C" size_t ga_gemm_splitk_workspace_bytes(int32_t M, int32_t N)
{
    if (M <= 0 || N <= 0) return 0;
    const size_t tiles = (size_t)((N + 127) / 128) * ((M + 191) / 192);
    return GA_GEMM_SPLITK_COUNTER_BYTES + tiles * 4 * 192 * 128 * 4;   // (two 96-row tiles x 4 splits fit the same bytes)
}

extern "C" int ga_gemm_bf16(const GaGemmArgs *a, void *stream)
{
    using namespace gadit;
    if (!a || !a->A || !a->W || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->K % BK != 0 || a->N % 4 != 0 || a->lda % 8 != 0 || a->ldo % 4 != 0 ||
        a->lda < a->K)
        return GA_DIT_ERR_BAD_SHAPE;
    if (a->epilogue == GA_GEMM_EPI_RESIDUAL && a->gate && a->rows_per_batch <= 0) return GA_DIT_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (a->vt && (a->epilogue != GA_GEMM_EPI_STORE_BF16 || a->vt_col0 % 64 != 0 || (a->N - a->vt_col0) % 64 != 0 ||
                  a->rows_per_batch <= 0 || a->vt_ld < a->rows_per_batch))
        return GA_DIT_ERR_BAD_SHAPE;
    if ((a->qk_cols0 || a->qk_cols1) &&
        (a->epilogue != GA_GEMM_EPI_STORE_BF16 || a->qk_cols0 % 64 || a->qk_cols1 % 64 || a->qk_cols1 < a->qk_cols0 ||
         a->qk_cols1 > a->N || (a->qk_cols0 && !a->qk_w0) || (a->qk_cols1 > a->qk_cols0 && !a->qk_w1)))
        return GA_DIT_ERR_BAD_SHAPE;
    if ((a->emit_x || a->emit_ss) && (a->epilogue != GA_GEMM_EPI_RESIDUAL || !a->emit_x || !a->emit_ss || a->N % 64 != 0 ||
                                      a->emit_ld % 8 != 0 || a->emit_ld < a->N))
        return GA_DIT_ERR_BAD_SHAPE;
    if (a->w_tiled && a->N % 8 != 0) return GA_DIT_ERR_BAD_SHAPE;
    if (a->row_ss && ((a->epilogue != GA_GEMM_EPI_STORE_BF16 && a->epilogue != GA_GEMM_EPI_GELU_BF16) || a->row_ss_tiles <= 0 ||
                      a->row_ss_tiles > 20 || a->row_ss_dim <= 0 || (uintptr_t)a->row_ss % 16 != 0))
        return GA_DIT_ERR_BAD_SHAPE;
    // (more than 16 partial sums per row: only the three-slot ring instances and the 2-slot general kernel read a fifth group)
    const bool wide_ss = a->row_ss && a->row_ss_tiles > 16;
    if ((a->emit_w || a->emit_scale) && (!a->emit_x || !a->emit_w || !a->emit_scale || a->rows_per_batch <= 0 || a->emit_scale_stride % 4 != 0))
        return GA_DIT_ERR_BAD_SHAPE;
    if (a->bias_stride && (!a->bias || a->rows_per_batch <= 0 || a->bias_stride % 4 != 0 || a->bias_stride < a->N)) return GA_DIT_ERR_BAD_SHAPE;
    if (a->k_rows && (a->epilogue != GA_GEMM_EPI_RESIDUAL || a->k_rows < 0 || a->k_rows > a->M)) return GA_DIT_ERR_BAD_SHAPE;
    GemmP p{a->M, a->N, a->K, a->rows_per_batch, a->A, a->W, a->bias, a->gate, a->out, a->lda, a->ldo, a->gate_stride,
                  a->vt, a->vt_col0, a->vt ? (a->N - a->vt_col0) / 64 : 0, a->vt_ld, a->qk_w0, a->qk_w1, a->qk_cols0,
                  a->qk_cols1, a->emit_x, a->emit_ss, a->emit_ld, a->row_ss, a->row_ss_tiles,
                  a->row_ss_dim > 0 ? 1.0f / (float)a->row_ss_dim : 0.f, a->row_ss_eps, a->w_tiled ? 1 : 0,
                  a->emit_w, a->emit_scale, a->emit_scale_stride, a->bias_stride, a->k_rows == a->M ? 0 : a->k_rows,
                  0, nullptr, nullptr, 0, 0};
    {   // write-through output stores (st16): same-box A/B (profiles/r6_wt_ab.txt) DiT-L at CFG batch 2 2.730 -> 2.691 ms per evaluation, DiT-B
        // 1.244 -> 1.234, batch 1 -0.3 %, CFG batch 4 +0.4 %: on for the one-round grids.  GA_GEMM_WT: 0 off, 1 always, 2 (default) M <= 2048
        static const int wt_env = [] { const char *e = getenv("GA_GEMM_WT"); return e ? atoi(e) : 2; }();
        p.wt = wt_env == 1 || (wt_env == 2 && a->M <= 2048);
    }
    // Tile / ring choice (256 CUs).  A workgroup tile is 128 weight rows x 32 MT activation rows (MT = 4, 3, 2, 1); its work is
    // proportional to MT plus a tile-independent share (prologue, weight tile, epilogue: about one MT unit, tools/gemm_sweep.py) and
    // the launch ends with the busiest CU, so the cost of a choice is ceil(workgroups / 256) * (MT + 1)
    // (two co-resident workgroups share their CU's LDS and matrix pipes: no faster than one after the other).  Ties go to the
    // larger tile (fewer operand bytes per flop).  Examples at N = 4096: M = 1536 -> 96-row tiles, 512 workgroups, exactly two
    // per CU (128-row tiles: 384 workgroups, half the CUs carry two of them: 10 units against 8); M = 768 -> 96-row tiles, 256
    // workgroups, exactly one per CU (128-row tiles: 192 workgroups of 5 units against 4).
    //   more than 256 workgroups -> 2-slot ring (<= 64 KiB): two workgroups per CU hide each other's latency
    //   otherwise                -> 4-slot ring: one workgroup per CU, three K-tiles of look-ahead
    // Round 3: the ring kernel where its K loop applies (K a multiple of 256, at least 512).  Tile by how many workgroups it puts
    // on the 256 CUs (tools/gemm_lab.hip, tools/gemm_sweep.py): 192 x 128 / 8 waves when that fills at least 5/8 of the chip
    // (fc1, qkv, everything at M >= 6144), else 96 x 64 (proj, fc2: 256 workgroups at M = 1536), else 64 x 64 (the M = 768
    // cross-attention projections).  The 96 x 64 tile has 32-column waves: for the per-head q/k norm its two waves exchange
    // their halves of the row sums of squares through LDS.
    // Round 6: deterministic split-K for the GEMMs whose output tiles cannot fill the chip (FusedMLP's second linear: K = 4 D; the wide
    // projections at 768 rows).  At M <= 1536 rows the chip is only filled by 96 x 64 / 64 x 64 tiles, which pull 1.3 MB of operands through
    // every CU's L1 miss path (the bound of these kernels); 192 x 128 tiles over a quarter of K each halve that.
    if (a->splitk_ws && p.k_rows == 0 && a->epilogue != GA_GEMM_EPI_STORE_F32) {
        const int sk_env = splitk_mode().load(std::memory_order_relaxed);
        const int nk = a->K / BK;
        const bool per_batch = a->bias_stride != 0 || a->emit_scale != nullptr;
        const bool rows48 = !per_batch || a->rows_per_batch % 48 == 0;
        const bool res = a->epilogue == GA_GEMM_EPI_RESIDUAL;
        const long long t192 = (long long)((a->N + 127) / 128) * ((a->M + 191) / 192), t96 = (long long)((a->N + 127) / 128) * ((a->M + 95) / 96);
        // 1: 192 x 128 x 4 splits (8 waves) | 2: 96 x 128 x 2 (4 waves) | 3: 96 x 128 x 4 | 4: 192 x 128 x 2.  The bf16-store epilogues
        // (qkv with its head norm / V^T, fc1 with GELU) only exist for configuration 4 -- what their shapes need at 768 rows.
        const bool ok4 = nk % 16 == 0 && nk >= 32, ok2 = nk % 8 == 0 && nk >= 16;
        auto fills = [](long long wgs) { return wgs >= 160 && wgs <= 256; };
        // MEASURED SLOWER on every DiT shape (profiles/r6_splitk.txt: fc2 at 1536 rows 25.5 -> 29.5 ... 35.0 us, at 768 rows 19.9 -> 22.8 ... 27.8,
        // DiT-L 3.11 -> 3.30 ms per evaluation): the memory-side hand-over costs more than the smaller K loop saves.  OFF unless asked for:
        // mode 6 = the shape rule below, 5 = the same for EPI 2 only, 1 ... 4 = one configuration.
        int cfg = 0;
        if ((sk_env == 5 || sk_env == 6) && rows48 && t192 < 160) {          // (a 192 x 128 grid that fills the chip on its own needs no split)
            if (ok2 && fills(t192 * 2)) cfg = 4;
            else if (res && ok4 && fills(t192 * 4)) cfg = 1;
            else if (res && ok4 && fills(t96 * 4)) cfg = 3;
            else if (res && ok2 && fills(t96 * 2)) cfg = 2;
        }
        if (sk_env == 1) cfg = res && rows48 && ok4 ? 1 : 0;
        else if (sk_env == 2) cfg = res && rows48 && ok2 ? 2 : 0;
        else if (sk_env == 3) cfg = res && rows48 && ok4 ? 3 : 0;
        else if (sk_env == 4) cfg = rows48 && ok2 ? 4 : 0;
        else if (sk_env == 5) cfg = res ? cfg : 0;          // by shape, residual GEMMs only
        const bool big = cfg == 1 || cfg == 4;
        const long long tiles = big ? t192 : t96;
        const int splits = (cfg == 2 || cfg == 4) ? 2 : 4;
        const size_t tile_bytes = (size_t)(big ? 192 : 96) * 128 * 4;
        if (cfg && tiles <= GA_GEMM_SPLITK_MAX_TILES &&
            (size_t)a->splitk_ws_bytes >= GA_GEMM_SPLITK_COUNTER_BYTES + (size_t)tiles * splits * tile_bytes && ((uintptr_t)a->splitk_ws & 255) == 0) {
            GemmP q = p;
            q.splits = splits;
            q.sk_count = static_cast<unsigned *>(a->splitk_ws);
            q.sk_part = reinterpret_cast<float *>(static_cast<char *>(a->splitk_ws) + GA_GEMM_SPLITK_COUNTER_BYTES);
            const dim3 gbig((unsigned)((a->N + 127) / 128), (unsigned)((a->M + 191) / 192), (unsigned)splits);
            const dim3 gmid((unsigned)((a->N + 127) / 128), (unsigned)((a->M + 95) / 96), (unsigned)splits);
            // (the LDS opt-in belongs to the function on one device; setting it is cheap and idempotent)
#define GA_SK_LAUNCH(KERNEL, GRID, THREADS, LDS)                                                                                   \
            do {                                                                                                                  \
                if (hipFuncSetAttribute((const void *)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return GA_DIT_ERR_LAUNCH; \
                hipLaunchKernelGGL(KERNEL, GRID, dim3(THREADS), LDS, s, q);                                                       \
            } while (0)
            if (cfg == 1) GA_SK_LAUNCH((gemm_ring_kernel<2, 4, 2, 3, 4, 4, 2, 4>), gbig, 512, 4 * 320 * BK * 2);
            else if (cfg == 2) GA_SK_LAUNCH((gemm_ring_kernel<2, 2, 2, 3, 4, 4, 2, 2>), gmid, 256, 4 * 224 * BK * 2);
            else if (cfg == 3) GA_SK_LAUNCH((gemm_ring_kernel<2, 2, 2, 3, 4, 4, 2, 4>), gmid, 256, 4 * 224 * BK * 2);
            else if (a->epilogue == GA_GEMM_EPI_RESIDUAL) GA_SK_LAUNCH((gemm_ring_kernel<2, 4, 2, 3, 4, 4, 2, 2>), gbig, 512, 4 * 320 * BK * 2);
            else if (a->epilogue == GA_GEMM_EPI_STORE_BF16) GA_SK_LAUNCH((gemm_ring_kernel<0, 4, 2, 3, 4, 4, 2, 2>), gbig, 512, 4 * 320 * BK * 2);
            else GA_SK_LAUNCH((gemm_ring_kernel<1, 4, 2, 3, 4, 4, 2, 2>), gbig, 512, 4 * 320 * BK * 2);
#undef GA_SK_LAUNCH
            return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
        }
    }
    {
        const int nk = a->K / BK;
        // (round 6: choosing the tile of a k_rows GEMM -- the cross-attention output projection of a CFG pair, half of whose rows carry no
        //  product -- by the rows that do measured slower, 2.927 -> 2.997 ms per DiT-L evaluation: not kept)
        const long long wg_big = (long long)((a->N + 127) / 128) * ((a->M + 191) / 192);
        const long long wg_mid = (long long)((a->N + 63) / 64) * ((a->M + 95) / 96);
        const long long wg_small = (long long)((a->N + 63) / 64) * ((a->M + 63) / 64);
        int ring = 0;
        // per-batch operands (bias rows, emit multipliers) are requested once per wave in the ring kernels: a wave's 48 (16) rows must
        // lie in one batch item
        const bool per_batch = a->bias_stride != 0 || a->emit_scale != nullptr;
        const bool rows48 = !per_batch || a->rows_per_batch % 48 == 0, rows16 = !per_batch || a->rows_per_batch % 16 == 0;
        // round 6: 96 x 128, four waves of 48 x 64 (the wave tile of the 192 x 128 kernel): for the wide projections at 768 rows, whose 96 x 64
        // grid is 1.5 - 2 workgroups per CU (qkv 384, fc1 512) while this one is 192 / 256 with 30 % fewer operand bytes on the busiest CU
        // same-box A/B (profiles/r6_ring4_ab.txt): DiT-L on the conditional sequence alone 2.376 -> 2.315 ms per evaluation, nothing else moves.  GA_GEMM_RING4=0: off
        static const int ring4_env = [] { const char *e = getenv("GA_GEMM_RING4"); return e ? atoi(e) : 1; }();
        // (a 48 x 64 two-wave tile that would give the N = 1024 residual GEMMs at 768 rows 256 workgroups instead of 192 measured 3 % slower per evaluation: not kept)
        const long long wg_96x128 = (long long)((a->N + 127) / 128) * ((a->M + 95) / 96);
        static const int ragged3_env = [] { const char *e = getenv("GA_GEMM_RAGGED3"); return e ? atoi(e) : 1; }();
        // round 6: K that is no multiple of 256 but one of 192 (DiT-PixArt-PCD-CLAY-XL: 1152, 4608) runs the same tiles on a ring of THREE slots
        // ... and, better (the same shapes ran 1.5 x longer on three slots: two tiles in flight per DMA latency instead of three): the FOUR-slot
        // ring with two remainder tiles in its peeled tail where K is 2 tiles past a multiple of 4 (1152 = 18 tiles); three slots for what is left
        // (GA_GEMM_REM=0: three slots as before, A/B aid)
        static const int rem_env = [] { const char *e = getenv("GA_GEMM_REM"); return e ? atoi(e) : 1; }();
        int nst_ring = 0, rem = 0;
        if (nk % 4 == 0 && nk >= 8) nst_ring = wide_ss ? 0 : 4;
        else if (rem_env && nk % 4 == 2 && nk >= 10) { nst_ring = 4; rem = 2; }
        else if (nk % 3 == 0 && nk >= 6) nst_ring = 3;
        const bool odd_k = nst_ring == 3 || rem != 0;     // (the instances added for such K: no fp32 store among them)
        if (nst_ring) {
            // (round 6: 144 instead of 160 -- DiT-B's qkv, 18 x 8 workgroups, is better off on this tile than on 864 of 64 x 64: 1.237 -> 1.222 ms per
            //  evaluation same-box; nothing else falls between the two.  GA_GEMM_BIGMIN: A/B aid)
            static const int bigmin = [] { const char *e = getenv("GA_GEMM_BIGMIN"); return e ? atoi(e) : 144; }();
            // (a 192 x 128 grid of a little over one round -- XL's fc1, 36 x 8 = 288 workgroups on 256 CUs -- pays two rounds for 1.1: 37.8 us
            //  against 28.2 on the 2-slot 128 x 128 kernel below, 432 workgroups two to a CU; the rule is confined to the three-slot shapes so
            //  that no released model's choice moves)
            static const int ragged_env = [] { const char *e = getenv("GA_GEMM_RAGGED"); return e ? atoi(e) : 1; }();
            const bool ragged_big = ragged_env && odd_k && wg_big > 256 && wg_big * 4 < ((wg_big + 255) / 256) * 256 * 3;
            if (odd_k && (a->epilogue == GA_GEMM_EPI_STORE_F32 || ragged_big)) ring = 0;   // (no such instance of the fp32 store: nothing asks for it)
            else if (wg_big >= bigmin && rows48) ring = 1;
            else if (ring4_env && a->epilogue != GA_GEMM_EPI_STORE_F32 && wg_96x128 >= 160 && wg_96x128 <= 256 && rows48) ring = 4;
            // (per-head norm on the 96 x 64 tile: its two 32-column waves exchange their sums through LDS, a barrier more than the
            //  64-column waves of the other tiles need -- worth it while the grid is one residency round, 2 x 256 workgroups:
            //  DiT-L's qkv at M = 768 13.4 -> 12.1 us; DiT-B's at M = 1536, 576 workgroups, is faster on 64 x 64, same-box A/B)
            // (round 6: a 64 x 128 four-wave tile for the residual GEMMs whose 96 x 64 grid is a little over one round -- width 1152 at 1536 rows:
            //  288 workgroups against 216 -- measured no better: fc2 37.6 us against 37.1, the K = 1152 projections 20.4 against 17.2: not kept)
            // round 6: a 96 x 64 grid of a little over one round (width 1152 at 1536 rows: 288 workgroups, two to a CU on 32 of them) goes to 64 x 64
            // tiles instead -- 432 workgroups, two to a CU everywhere: the busiest CU pulls 2 x 16 KB per K-tile instead of 2 x 20.  Same-box, XL per
            // evaluation 5.07 -> 4.78 ms.  No released model's shape meets the condition.  GA_GEMM_RAGGED3=0: off (A/B aid)
            else if (ragged3_env && a->epilogue == GA_GEMM_EPI_RESIDUAL && wg_mid > 256 && wg_mid * 4 < ((wg_mid + 255) / 256) * 256 * 3 &&
                     wg_small <= 512 && rows16) ring = 3;
            // (round 6: a 192 x 64 eight-wave tile for the residual GEMMs at 3072 rows -- 256 workgroups, one to a CU, instead of 512 of 96 x 64, two to
            //  a CU: 32 KB per K-tile through a CU instead of 2 x 20 -- measured the same, DiT-L at CFG batch 4 5.03 / 5.03 ms per evaluation: not kept)
            else if (wg_mid >= 160 && rows48 && (!(a->qk_cols0 || a->qk_cols1) || wg_mid <= 512)) ring = 2;
            else if (wg_small >= 96 && rows16) ring = 3;
        }
#ifdef GA_TUNING  // tuning builds only: GA_GEMM_RING = 0 old kernel, 1 / 2 / 3 force a ring tile
        if (const char *e = getenv("GA_GEMM_RING")) { const int c = atoi(e); if (c == 0 || (nk % 4 == 0 && nk >= 8 && c >= 1 && c <= 3)) ring = c; }
#endif
        // (round 6: a 96-row x 192-column six-wave tile that fills all 256 CUs for DiT-L's qkv -- 16 x 16 workgroups instead of the 24 x 8
        //  = 192 of the 192 x 128 tile, 36 KB per K-tile instead of 40 -- measured the same to the microsecond, same-box A/B
        //  2.72 / 2.72 ms per evaluation: not kept)
        if (ring) {
            // (the LDS opt-in belongs to the function ON ONE DEVICE: one flag per device, set by whichever thread gets there -- idempotent)
            static std::atomic<bool> ring_attr_dev[64];
            int dev_ = 0;
            (void)hipGetDevice(&dev_);
            std::atomic<bool> &ring_attr_set = ring_attr_dev[dev_ >= 0 && dev_ < 64 ? dev_ : 63];
            if (!ring_attr_set.load(std::memory_order_acquire) || dev_ >= 64) {
#define GA_RATTR(E)                                                                                                       \
                (void)hipFuncSetAttribute((const void *)gemm_ring_kernel<E, 4, 2, 3, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 320 * BK * 2); \
                (void)hipFuncSetAttribute((const void *)gemm_ring_kernel<E, 2, 2, 3, 2, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 160 * BK * 2); \
                (void)hipFuncSetAttribute((const void *)gemm_ring_kernel<E, 4, 1, 1, 4, 4, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * BK * 2);
                GA_RATTR(0) GA_RATTR(1) GA_RATTR(2) GA_RATTR(3)
#undef GA_RATTR
                ring_attr_set.store(true, std::memory_order_release);
            }
            // tile -> XCD blocking for the residual GEMMs on the 96 x 64 / 64 x 64 tiles (GemmP.xmap).  Same-box A/B (profiles/r6_xmap_ab.txt):
            // it pays where the natural dealing is ragged or the grid is more than one round -- DiT-B (12 tile columns: an XCD's
            // workgroups share neither weights nor activations systematically) 1.267 -> 1.229 ms per evaluation, DiT-L at CFG batch 4
            // (512 workgroups) 5.21 -> 5.10 -- and costs 1 % where every XCD already owns two whole tile columns of a one-round grid
            // (DiT-L at CFG batch 2: 2.930 -> 2.960; batch 1: 2.409 -> 2.422).  GA_GEMM_XMAP: 0 off, 1 always (where the grid divides), 2 (default) by that rule
            static const int xmap_env = [] { const char *e = getenv("GA_GEMM_XMAP"); return e ? atoi(e) : 2; }();
            GemmP pr = p;
            {
                const unsigned gx = ring == 2 ? (unsigned)((a->N + 63) / 64) : (ring == 3 ? (unsigned)((a->N + 63) / 64) : 1u);
                const unsigned gy = ring == 2 ? (unsigned)((a->M + 95) / 96) : (ring == 3 ? (unsigned)((a->M + 63) / 64) : 1u);
                const bool pays = xmap_env == 1 || (xmap_env == 2 && (gx % 8 != 0 || gx * gy > 256));
                if (pays && ring != 1 && a->epilogue == GA_GEMM_EPI_RESIDUAL && gx % 2 == 0 && gy % 4 == 0 && (gx * gy) % 8 == 0) pr.xmap = 1;
            }
            // (round 6: the k-step software pipeline of the 192 x 128 kernel on the 96 x 64 tile measured 1.5 - 2 % slower per evaluation: not kept)
#define GA_RLAUNCH_N(E, NSTV, RV)                                                                                         \
            if (ring == 4) {                                                                                              \
                if (hipFuncSetAttribute((const void *)gemm_ring_kernel<E, 2, 2, 3, 4, NSTV, 2, 0, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTV * 224 * BK * 2) != hipSuccess) return GA_DIT_ERR_LAUNCH; \
                hipLaunchKernelGGL((gemm_ring_kernel<E, 2, 2, 3, 4, NSTV, 2, 0, RV>), dim3((unsigned)((a->N + 127) / 128), (unsigned)((a->M + 95) / 96)), \
                                   dim3(256), NSTV * 224 * BK * 2, s, p);                                                 \
            } else if (ring == 1) {                                                                                       \
                if ((NSTV != 4 || RV != 0) && hipFuncSetAttribute((const void *)gemm_ring_kernel<E, 4, 2, 3, 4, NSTV, 2, 0, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTV * 320 * BK * 2) != hipSuccess) return GA_DIT_ERR_LAUNCH; \
                hipLaunchKernelGGL((gemm_ring_kernel<E, 4, 2, 3, 4, NSTV, 2, 0, RV>), dim3((unsigned)((a->N + 127) / 128), (unsigned)((a->M + 191) / 192)), \
                                   dim3(512), NSTV * 320 * BK * 2, s, p);                                                 \
            } else if (ring == 2) {                                                                                       \
                if ((NSTV != 4 || RV != 0) && hipFuncSetAttribute((const void *)gemm_ring_kernel<E, 2, 2, 3, 2, NSTV, 0, 0, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTV * 160 * BK * 2) != hipSuccess) return GA_DIT_ERR_LAUNCH; \
                hipLaunchKernelGGL((gemm_ring_kernel<E, 2, 2, 3, 2, NSTV, 0, 0, RV>), dim3((unsigned)((a->N + 63) / 64), (unsigned)((a->M + 95) / 96)), \
                                   dim3(256), NSTV * 160 * BK * 2, s, pr);                                                \
            } else {                                                                                                      \
                if ((NSTV != 4 || RV != 0) && hipFuncSetAttribute((const void *)gemm_ring_kernel<E, 4, 1, 1, 4, NSTV, 0, 0, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, NSTV * 128 * BK * 2) != hipSuccess) return GA_DIT_ERR_LAUNCH; \
                hipLaunchKernelGGL((gemm_ring_kernel<E, 4, 1, 1, 4, NSTV, 0, 0, RV>), dim3((unsigned)((a->N + 63) / 64), (unsigned)((a->M + 63) / 64)), \
                                   dim3(256), NSTV * 128 * BK * 2, s, pr);                                                \
            }
            // (round 6: a deeper ring -- 6 or 8 slots of the 96 x 64 tile, 120 / 160 KiB -- for the residual GEMMs that run one workgroup per CU
            //  measured no gain, DiT-L 2.79 -> 2.81 / 2.83 ms per evaluation same-box: unlike the 192 x 128 tile at K = 1152, whose step from three
            //  slots to four was worth 1.27 x, these are not bound by the tiles in flight)
#define GA_RLAUNCH(E) if (nst_ring == 4 && rem == 0) { GA_RLAUNCH_N(E, 4, 0) } else if (nst_ring == 4) { GA_RLAUNCH_N(E, 4, 2) } else { GA_RLAUNCH_N(E, 3, 0) }
            switch (a->epilogue) {
            case GA_GEMM_EPI_STORE_BF16: GA_RLAUNCH(0) break;
            case GA_GEMM_EPI_GELU_BF16: GA_RLAUNCH(1) break;
            case GA_GEMM_EPI_RESIDUAL: GA_RLAUNCH(2) break;
            case GA_GEMM_EPI_STORE_F32: GA_RLAUNCH_N(3, 4, 0) break;     // (odd_k never gets here)
            default: return GA_DIT_ERR_BAD_SHAPE;
            }
#undef GA_RLAUNCH_N
#undef GA_RLAUNCH
            return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
        }
    }
    const long long ncols = (a->N + BN - 1) / BN;
    int mt = 4;
    long long best = -1;
    for (int cand = 4; cand >= 1; --cand) {
        const long long wgs = ncols * ((a->M + 32 * cand - 1) / (32 * cand));
        const long long cost = ((wgs + 255) / 256) * (cand + 1);
        if (best < 0 || cost < best) { best = cost; mt = cand; }
    }
    long long wgs = ncols * ((a->M + 32 * mt - 1) / (32 * mt));
    int nst = (wgs > 256 || wide_ss) ? 2 : 4;
#ifdef GA_TUNING  // tuning builds only: GA_GEMM_CFG = 10 * MT + ring slots
    if (const char *e = getenv("GA_GEMM_CFG")) { const int c = atoi(e); if (c / 10 >= 1 && c / 10 <= 4 && (c % 10 == 2 || c % 10 == 4)) { mt = c / 10; nst = c % 10; } }
#endif
    static std::atomic<bool> attr_dev[64];
    int dev2_ = 0;
    (void)hipGetDevice(&dev2_);
    std::atomic<bool> &attr_set = attr_dev[dev2_ >= 0 && dev2_ < 64 ? dev2_ : 63];
    if (!attr_set.load(std::memory_order_acquire) || dev2_ >= 64) {  // > 64 KiB of dynamic LDS has to be opted into once per kernel and device
#define GA_ATTR(E)                                                                                                  \
        (void)hipFuncSetAttribute((const void *)gemm_bf16_kernel<E, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  4 * (BN + 128) * BK * 2);                                                          \
        (void)hipFuncSetAttribute((const void *)gemm_bf16_kernel<E, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  4 * (BN + 96) * BK * 2);                                                           \
        (void)hipFuncSetAttribute((const void *)gemm_bf16_kernel<E, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  4 * (BN + 64) * BK * 2);                                                           \
        (void)hipFuncSetAttribute((const void *)gemm_bf16_kernel<E, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  4 * (BN + 32) * BK * 2);
        GA_ATTR(0) GA_ATTR(1) GA_ATTR(2) GA_ATTR(3)
#undef GA_ATTR
        attr_set.store(true, std::memory_order_release);
    }
#define GA_LAUNCH_MT(E, NSTV, MTV)                                                                                   \
    hipLaunchKernelGGL((gemm_bf16_kernel<E, NSTV, MTV>), dim3((unsigned)ncols, (unsigned)((a->M + 32 * MTV - 1) / (32 * MTV))), \
                       dim3(256), NSTV * (BN + 32 * MTV) * BK * 2, s, p)
#define GA_LAUNCH(E)                                                                                                 \
    if (nst == 2) {                                                                                                  \
        if (mt == 4) GA_LAUNCH_MT(E, 2, 4); else if (mt == 3) GA_LAUNCH_MT(E, 2, 3);                                  \
        else if (mt == 2) GA_LAUNCH_MT(E, 2, 2); else GA_LAUNCH_MT(E, 2, 1);                                          \
    } else {                                                                                                         \
        if (mt == 4) GA_LAUNCH_MT(E, 4, 4); else if (mt == 3) GA_LAUNCH_MT(E, 4, 3);                                  \
        else if (mt == 2) GA_LAUNCH_MT(E, 4, 2); else GA_LAUNCH_MT(E, 4, 1);                                          \
    }
    switch (a->epilogue) {
    case GA_GEMM_EPI_STORE_BF16: GA_LAUNCH(0) break;
    case GA_GEMM_EPI_GELU_BF16: GA_LAUNCH(1) break;
    case GA_GEMM_EPI_RESIDUAL: GA_LAUNCH(2) break;
    case GA_GEMM_EPI_STORE_F32: GA_LAUNCH(3) break;
    default: return GA_DIT_ERR_BAD_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}
