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
// output.  Workgroup tile 128(n) x 128(m) x 64(k), 4 waves as 2x2, each wave 4x4 fragments (64 accumulator VGPRs);
// global -> VGPR -> LDS staging one K-tile ahead (16-byte loads), LDS rows padded to 144 B so the 16 rows a fragment
// read touches fall into 16 different bank groups.
#include "dit_common.h"

namespace gadit {

constexpr int BM = 128, BN = 128, BK = 64, LDS_LD = BK + 8;  // leading dimension in bf16 elements (144 bytes)

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GaGemmArgs a)
{
    __shared__ __attribute__((aligned(16))) uint16_t sW[BN * LDS_LD];
    __shared__ __attribute__((aligned(16))) uint16_t sA[BM * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int M = a.M, N = a.N, K = a.K;

    // staging assignment: 1024 16-byte chunks per operand tile, 4 per thread
    int srow[4], scol[4];
    const uint16_t *gW[4], *gA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + 256 * i;
        srow[i] = c >> 3;
        scol[i] = (c & 7) * 8;
        const int rn = min(n0 + srow[i], N - 1), rm = min(m0 + srow[i], M - 1);
        gW[i] = a.W + (size_t)rn * K + scol[i];
        gA[i] = a.A + (size_t)rm * a.lda + scol[i];
    }
    uint4 rW[4], rA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rW[i] = *reinterpret_cast<const uint4 *>(gW[i]);
        rA[i] = *reinterpret_cast<const uint4 *>(gA[i]);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = (lane >> 4) * 8;
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4 *>(&sW[srow[i] * LDS_LD + scol[i]]) = rW[i];
            *reinterpret_cast<uint4 *>(&sA[srow[i] * LDS_LD + scol[i]]) = rA[i];
        }
        __syncthreads();
        if (kt + 1 < nk) {  // next K-tile in flight while this one is multiplied
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rW[i] = *reinterpret_cast<const uint4 *>(gW[i] + (size_t)(kt + 1) * BK);
                rA[i] = *reinterpret_cast<const uint4 *>(gA[i] + (size_t)(kt + 1) * BK);
            }
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            bf16x8 fw[4], fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fw[i] = *reinterpret_cast<const bf16x8 *>(&sW[(wn * 64 + i * 16 + frow) * LDS_LD + kk + fk]);
                fa[i] = *reinterpret_cast<const bf16x8 *>(&sA[(wm * 64 + i * 16 + frow) * LDS_LD + kk + fk]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: lane holds acc[i][j][r] = C[m = m0 + wm*64 + j*16 + (lane&15)][n = n0 + wn*64 + i*16 + (lane>>4)*4 + r]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + (lane & 15);
        if (m >= M) continue;
        const float *gate_row = nullptr;
        if (EPI == GA_GEMM_EPI_RESIDUAL && a.gate) gate_row = a.gate + (size_t)(m / a.rows_per_batch) * a.gate_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4;
            if (n >= N) continue;
            f32x4 v = acc[i][j];
            if (a.bias) {
                const float4 b = *reinterpret_cast<const float4 *>(a.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (EPI == GA_GEMM_EPI_GELU_BF16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752f));
            }
            if (EPI == GA_GEMM_EPI_STORE_BF16 || EPI == GA_GEMM_EPI_GELU_BF16) {
                uint2 p = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                *reinterpret_cast<uint2 *>(static_cast<uint16_t *>(a.out) + (size_t)m * a.ldo + n) = p;
            } else {
                float4 *dst = reinterpret_cast<float4 *>(static_cast<float *>(a.out) + (size_t)m * a.ldo + n);
                if (EPI == GA_GEMM_EPI_RESIDUAL) {
                    float4 g = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (gate_row) g = *reinterpret_cast<const float4 *>(gate_row + n);
                    const float4 x = *dst;
                    *dst = make_float4(x.x + g.x * v[0], x.y + g.y * v[1], x.z + g.z * v[2], x.w + g.w * v[3]);
                } else {
                    *dst = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

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
    const dim3 grid((a->N + BN - 1) / BN, (a->M + BM - 1) / BM);
    switch (a->epilogue) {
    case GA_GEMM_EPI_STORE_BF16: hipLaunchKernelGGL(gemm_bf16_kernel<0>, grid, dim3(256), 0, s, *a); break;
    case GA_GEMM_EPI_GELU_BF16: hipLaunchKernelGGL(gemm_bf16_kernel<1>, grid, dim3(256), 0, s, *a); break;
    case GA_GEMM_EPI_RESIDUAL: hipLaunchKernelGGL(gemm_bf16_kernel<2>, grid, dim3(256), 0, s, *a); break;
    case GA_GEMM_EPI_STORE_F32: hipLaunchKernelGGL(gemm_bf16_kernel<3>, grid, dim3(256), 0, s, *a); break;
    default: return GA_DIT_ERR_BAD_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}
