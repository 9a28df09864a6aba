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

#include "dit_common.h"

namespace gadit {

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
};

__device__ __forceinline__ void glds16(const uint16_t *gsrc, uint16_t *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int EPI, int MT>
__device__ __forceinline__ void gemm_epilogue(const GemmP &p, f32x4 (&acc)[4][MT], int m0, int n0, int wn, int wm, int lane)
{
    const int M = p.M, N = p.N;
    // epilogue: lane holds acc[i][j][r] = C[m = m0 + wm*MT*16 + j*16 + (lane&15)][n = n0 + wn*64 + i*16 + (lane>>4)*4 + r];
    // the 64 columns of a wave are one attention head: its 64 values of row m sit in 4 fragments x 4 lane-groups x 4
    // registers, so the per-head RMSNorm is an in-lane sum plus two xor-shuffles
    const int nhead = n0 + wn * 64;
    const float *qkw = nullptr;
    if (EPI == GA_GEMM_EPI_STORE_BF16) {
        if (nhead < p.qk_cols0) qkw = p.qk_w0;
        else if (nhead < p.qk_cols1) qkw = p.qk_w1;
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * (MT * 16) + j * 16 + (lane & 15);
        const float *gate_row = nullptr;
        if (EPI == GA_GEMM_EPI_RESIDUAL && p.gate && m < M) gate_row = p.gate + (size_t)(m / p.rows_per_batch) * p.gate_stride;
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nhead + i * 16 + (lane >> 4) * 4;
            v[i] = acc[i][j];
            if (p.bias && n < N) {
                const float4 b = *reinterpret_cast<const float4 *>(p.bias + n);
                v[i][0] += b.x; v[i][1] += b.y; v[i][2] += b.z; v[i][3] += b.w;
            }
        }
        if (EPI == GA_GEMM_EPI_STORE_BF16 && qkw) {  // wave-uniform
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            const float rs = rsqrtf(ss * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 w = *reinterpret_cast<const float4 *>(qkw + i * 16 + (lane >> 4) * 4);
                v[i][0] *= rs * w.x; v[i][1] *= rs * w.y; v[i][2] *= rs * w.z; v[i][3] *= rs * w.w;
            }
        }
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nhead + i * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            if (EPI == GA_GEMM_EPI_GELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i][r] = 0.5f * v[i][r] * (1.0f + erff(v[i][r] * 0.70710678118654752f));
            }
            if (EPI == GA_GEMM_EPI_STORE_BF16 && p.vt && n >= p.vt_col0) {
                // V projection: write V^T[(b*heads + h)*64 + d][token]; 16 consecutive lanes hold 16 consecutive tokens
                const int b = m / p.rows_per_batch, tok = m - b * p.rows_per_batch, dn = n - p.vt_col0;
                uint16_t *dst = p.vt + ((size_t)b * p.heads * 64 + dn) * p.vt_ld + tok;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(size_t)r * p.vt_ld] = f32_to_bf16(v[i][r]);
            } else if (EPI == GA_GEMM_EPI_STORE_BF16 || EPI == GA_GEMM_EPI_GELU_BF16) {
                uint2 pk = make_uint2(pack_bf16x2(v[i][0], v[i][1]), pack_bf16x2(v[i][2], v[i][3]));
                *reinterpret_cast<uint2 *>(static_cast<uint16_t *>(p.out) + (size_t)m * p.ldo + n) = pk;
            } else {
                float4 *dst = reinterpret_cast<float4 *>(static_cast<float *>(p.out) + (size_t)m * p.ldo + n);
                if (EPI == GA_GEMM_EPI_RESIDUAL) {
                    float4 gt = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (gate_row) gt = *reinterpret_cast<const float4 *>(gate_row + n);
                    const float4 x = *dst;
                    *dst = make_float4(x.x + gt.x * v[i][0], x.y + gt.y * v[i][1], x.z + gt.z * v[i][2], x.w + gt.w * v[i][3]);
                } else {
                    *dst = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
                }
            }
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
        srcW[i] = p.W + (size_t)min(n0 + row, N - 1) * K + ((lane & 7) ^ (row & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = (wave * MT + i) * 8 + (lane >> 3);
        srcA[i] = p.A + (size_t)min(m0 + row, M - 1) * p.lda + ((lane & 7) ^ (row & 7)) * 8;
    }
    f32x4 acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, g = lane >> 4;
    const int nk = K / BK;

    // Buffer indices are compile-time constants in every use below (the K loop is unrolled by two): with a run-time
    // buffer index hipcc cannot separate the DMA destination from the fragment reads and drains the DMA (vmcnt(0))
    // in front of every ds_read, which serialises load and math.
#define GA_STAGE(BUF, KT)                                                                             \
    do {                                                                                              \
        uint16_t *bw_ = smem + (BUF) * SLOT, *ba_ = bw_ + TILE_ELEMS;                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
            glds16(srcW[i] + (size_t)(KT) * BK, bw_ + (wave * 4 + i) * 8 * BK);                       \
        _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                \
            glds16(srcA[i] + (size_t)(KT) * BK, ba_ + (wave * MT + i) * 8 * BK);                      \
    } while (0)
#define GA_COMPUTE(BUF)                                                                               \
    do {                                                                                              \
        const uint16_t *bw_ = smem + (BUF) * SLOT, *ba_ = bw_ + TILE_ELEMS;                            \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                            \
            bf16x8 fw[4], fa[MT];                                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                const int rw = wn * 64 + i * 16 + frow;                                               \
                fw[i] = *reinterpret_cast<const bf16x8 *>(bw_ + rw * BK + (((kk * 4 + g) ^ (rw & 7)) * 8)); \
            }                                                                                         \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                          \
                const int ra = wm * (MT * 16) + i * 16 + frow;                                        \
                fa[i] = *reinterpret_cast<const bf16x8 *>(ba_ + ra * BK + (((kk * 4 + g) ^ (ra & 7)) * 8)); \
            }                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
                _Pragma("unroll") for (int j = 0; j < MT; ++j)                                        \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0); \
        }                                                                                             \
    } while (0)

    // ring of NST slots: tiles kt+1 .. kt+NST-1 are in flight while tile kt is multiplied (NST-1 tiles of look-ahead).
    // NST = 4 (128 KiB, one workgroup per CU) for the small grids, NST = 2 (64 KiB, two workgroups per CU, which hide each
    // other's latency) when the grid has more than one workgroup per CU -- chosen by the host from the grid size.
#define GA_WAIT_TILES_IN_FLIGHT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((N) * DMA_PER_TILE) : "memory")
#define GA_PHASE(BUF, KT)                                                                              \
    do {                                                                                               \
        const int rem_ = nk - 1 - (KT); /* tiles after KT; at most NST-2 of them are issued so far */    \
        if (NST >= 4 && rem_ >= 2) GA_WAIT_TILES_IN_FLIGHT(2);                                          \
        else if (NST >= 3 && rem_ >= 1) GA_WAIT_TILES_IN_FLIGHT(1);                                     \
        else GA_WAIT_TILES_IN_FLIGHT(0);                                                                \
        __builtin_amdgcn_s_barrier(); /* everyone's part of tile KT landed; everyone left tile KT-1 */  \
        if ((KT) + NST - 1 < nk) GA_STAGE(((BUF) + NST - 1) % NST, (KT) + NST - 1);                     \
        GA_COMPUTE(BUF);                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* my fragment reads of this buffer are done */ \
    } while (0)

    GA_STAGE(0, 0);
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

    gemm_epilogue<EPI, MT>(p, acc, m0, n0, wn, wm, lane);
}

// A register-FIFO variant of this kernel (global_load_dwordx4 into D = 4 / 8 K-tiles of VGPRs, ds_write into a double
// buffer) was built and measured in round 1 to test whether more bytes in flight would lift the small-M GEMMs: it does
// not (N = 4096: 46 us vs 29 us; deeper FIFOs slower still).  PMC (tools/pmc_gemm.sh): L1->L2 read latency averages
// 320 cycles and the L1 streams ~25 B/clk/CU, far below its 64 B/clk port -- the miss path of the L1, not the prefetch
// depth, caps these tiles, so only more reuse per byte (larger workgroup tiles on grids that still fill the chip) helps.

}  // namespace gadit

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
    const GemmP p{a->M, a->N, a->K, a->rows_per_batch, a->A, a->W, a->bias, a->gate, a->out, a->lda, a->ldo, a->gate_stride,
                  a->vt, a->vt_col0, a->vt ? (a->N - a->vt_col0) / 64 : 0, a->vt_ld, a->qk_w0, a->qk_w1, a->qk_cols0,
                  a->qk_cols1};
    // Tile / ring choice from the grid size (256 CUs):
    //   > 256 workgroups of 128x128          -> 128-row tiles, 2-slot ring (64 KiB): two workgroups share a CU
    //   <= 256 of them, but > 128             -> 128-row tiles, 4-slot ring (128 KiB): one workgroup per CU, deep look-ahead
    //   <= 128 (the N = 1024 GEMMs at M=1536) -> 64-row tiles, 4-slot ring (96 KiB): twice the workgroups
    const long long wg128 = (long long)((a->N + BN - 1) / BN) * ((a->M + 127) / 128);
    int cfg = wg128 > 256 ? 0 : (wg128 > 128 ? 1 : 2);
    if (const char *e = getenv("GA_GEMM_CFG")) cfg = atoi(e) % 3;  // tuning aid
    static bool attr_set = false;
    if (!attr_set) {  // > 64 KiB of dynamic LDS has to be opted into once per kernel
#define GA_ATTR(E)                                                                                                  \
        (void)hipFuncSetAttribute((const void *)gemm_bf16_kernel<E, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  4 * (BN + 128) * BK * 2);                                                          \
        (void)hipFuncSetAttribute((const void *)gemm_bf16_kernel<E, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  4 * (BN + 64) * BK * 2);
        GA_ATTR(0) GA_ATTR(1) GA_ATTR(2) GA_ATTR(3)
#undef GA_ATTR
        attr_set = true;
    }
#define GA_LAUNCH(E)                                                                                               \
    if (cfg == 0)                                                                                                  \
        hipLaunchKernelGGL((gemm_bf16_kernel<E, 2, 4>), dim3((a->N + BN - 1) / BN, (a->M + 127) / 128), dim3(256),  \
                           2 * (BN + 128) * BK * 2, s, p);                                                          \
    else if (cfg == 1)                                                                                             \
        hipLaunchKernelGGL((gemm_bf16_kernel<E, 4, 4>), dim3((a->N + BN - 1) / BN, (a->M + 127) / 128), dim3(256),  \
                           4 * (BN + 128) * BK * 2, s, p);                                                          \
    else                                                                                                           \
        hipLaunchKernelGGL((gemm_bf16_kernel<E, 4, 2>), dim3((a->N + BN - 1) / BN, (a->M + 63) / 64), dim3(256),    \
                           4 * (BN + 64) * BK * 2, s, p);
    switch (a->epilogue) {
    case GA_GEMM_EPI_STORE_BF16: GA_LAUNCH(0) break;
    case GA_GEMM_EPI_GELU_BF16: GA_LAUNCH(1) break;
    case GA_GEMM_EPI_RESIDUAL: GA_LAUNCH(2) break;
    case GA_GEMM_EPI_STORE_F32: GA_LAUNCH(3) break;
    default: return GA_DIT_ERR_BAD_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}
