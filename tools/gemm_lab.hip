// tools/gemm_lab.hip -- standalone test bed for the K loop / tile geometry of the DiT bf16 GEMM (gfx950).
//
//   C[m][n] = sum_k A[m][k] * W[n][k] + bias[n]   (A [M,K], W [N,K] bf16 K-contiguous, fp32 accumulate, bf16 store)
//
// One templated kernel, many instantiations, one process: every variant is checked against a naive fp32 kernel on the
// full output and timed with HIP events on the launch stream, with the weight operand rotated through more copies than
// the Infinity Cache holds ("cold": what a DiT evaluation sees -- 600 MB of weights per evaluation) and with one copy
// ("warm").  Nothing here is product code; the winning structure moves into gaussiananything_amd/csrc/dit_gemm.hip.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_build/gemm_lab tools/gemm_lab.hip
// run (GPU box): tools/_build/gemm_lab [filter-substring]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

struct P {
    int M, N, K;
    const uint16_t *A, *W;
    const float *bias;
    uint16_t *out;
    long long lda, ldo, ldw;
    int wtiled;   // 1: W stored as [N/8][K/64][8][64] (every DMA instruction reads 1 KiB of consecutive bytes)
    int xmap;     // 1: XCD-contiguous tile ids (see lab_kernel)
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

__device__ __forceinline__ float gelu_erf(float v)
{
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
    const float erf_abs = 1.0f - poly * t * e;
    return 0.5f * (v + fabsf(v) * erf_abs);
}

__device__ __forceinline__ void glds16(const uint16_t *gsrc, uint16_t *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int MF> struct AccT;
template <> struct AccT<16> { using T = f32x4; };
template <> struct AccT<32> { using T = f32x16; };

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// MF: MFMA edge (16: v_mfma_f32_16x16x32_bf16, 32: v_mfma_f32_32x32x16_bf16); WM x WN waves; FM x FN fragments per wave;
// NST ring slots; PIPE 0: all fragment reads of a K-tile, then its MFMAs; 1: k-step software pipeline (fragments of the next
// k-step requested before the MFMAs of the current one, tile hand-over inside the last k-step); EPI 0 bias, 1 bias + GELU
// WK (round 4, PIPE 0 / MF 16 only): WK wave groups share the tile, group wk takes the k-steps wk * KS/WK ... of every K-tile (same LDS
// traffic, twice the waves per SIMD to hide ds_read -> MFMA latency); the groups' accumulators meet in LDS after the K loop
template <int MF, int WM, int WN, int FM, int FN, int NST, int PIPE, int EPI, int R, int ABL = 0, int WK = 1>
__global__ __launch_bounds__(64 * WM * WN * WK) void lab_kernel(P p)
{
    using acc_t = typename AccT<MF>::T;
    static_assert(WK == 1 || (PIPE == 0 && MF == 16 && EPI < 2), "the k-step split is built for the batch-read loop");
    constexpr int NW = WM * WN * WK, BM = WM * FM * MF, BN = WN * FN * MF, BK = 64;
    constexpr int KS = MF == 16 ? 2 : 4;   // k-steps per K-tile
    constexpr int CPK = 8 / KS;            // 16-byte chunks per k-step
    constexpr int ROWS = BN + BM;
    // UNEVEN (WK > 1 only): ROWS / 8 row groups do not divide among the waves -- the first (ROWS / 8) % NW waves issue DPT instructions per
    // K-tile, the others DPT - 1, each group with its own counted waits
    constexpr bool UNEVEN = (ROWS / 8) % NW != 0;
    static_assert(!UNEVEN || (WK > 1 && PIPE == 0), "DMA rows must divide among the waves");
    constexpr int DPT = (ROWS / 8 + NW - 1) / NW;     // DMA instructions per wave per K-tile (at most)
    constexpr int SLOT = ROWS * BK;        // elements per ring slot
    static_assert((NST - 2) * DPT <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / (WM * WN), w2 = wave % (WM * WN);
    const int wn = w2 / WM, wm = w2 % WM;
    // LAB_XMAP (p.xmap): workgroups are dealt to the 8 XCDs round-robin by their linear id; remapped so that every XCD owns a CONTIGUOUS
    // range of tile ids (x fastest: whole rows of tiles, i.e. all of W and 1/8 of A per XCD's L2 instead of 1/8 of W and all of A)
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xmap) {
        const int gx = gridDim.x, total = gx * gridDim.y, id = by * gx + bx;
        if (total % 8 == 0) { const int nid = (id & 7) * (total >> 3) + (id >> 3); bx = nid % gx; by = nid / gx; }
    }
    const int n0 = bx * BN, m0 = by * BM;
    const int M = p.M, N = p.N, K = p.K;
    // ABL (timing-only ablations, wrong results): 1 no output stores, 2 every workgroup reads the A rows of tile 0, 4 ... the W rows
    // of tile 0, 8 no MFMAs, 16 no DMA after the prologue, 32 only the last NST + R K-tiles
    const int nk = (ABL & 32) ? NST + R : K / BK;
    const int n0w = (ABL & 4) ? 0 : n0, m0a = (ABL & 2) ? 0 : m0;

    // DMA sources: instruction i of this wave fills rows q*8 .. q*8+7 of the slot image [W rows | A rows], q = i*NW + wave
    const uint16_t *src[DPT];
    int kstep[DPT];
    const bool full = !UNEVEN || ((DPT - 1) * NW + wave) < ROWS / 8;   // wave-uniform: this wave issues all DPT instructions
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
        const int row = min((i * NW + wave) * 8 + (lane >> 3), ROWS - 1);
        kstep[i] = (row < BN && p.wtiled) ? 512 : BK;
        if (row < BN) {
            const int sw = MF == 16 ? (row & 7) : ((row >> 1) & 7);
            if (p.wtiled) src[i] = p.W + (size_t)((n0w + row) >> 3) * (K / 64) * 512 + (row & 7) * 64 + ((lane & 7) ^ sw) * 8;
            else src[i] = p.W + (size_t)min(n0w + row, N - 1) * p.ldw + ((lane & 7) ^ sw) * 8;
        } else {
            const int r = row - BN;
            const int sw = MF == 16 ? (r & 7) : ((r >> 1) & 7);
            src[i] = p.A + (size_t)min(m0a + r, M - 1) * p.lda + ((lane & 7) ^ sw) * 8;
        }
    }
    acc_t acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < (MF == 16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

    const int lrow = lane & (MF - 1), lg = lane / MF;
    const int lsw = MF == 16 ? (lrow & 7) : ((lrow >> 1) & 7);
    // element offset of this lane's 16-byte piece inside a row-major [rows][64] tile, k-step 0: row*64 + ((lg ^ lsw) * 8);
    // k-step ks flips the chunk bits by ks*CPK
    const int lane_off = lrow * BK + ((lg ^ lsw) * 8);

    auto stage = [&](auto bufc, int kt) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        uint16_t *base = smem + BUF * SLOT;
        if ((ABL & 16) && kt >= NST - 1) return;
#pragma unroll
        for (int i = 0; i < DPT; ++i)
            if (i + 1 < DPT || full) glds16(src[i] + (size_t)kt * kstep[i], base + (i * NW + wave) * 8 * BK);
    };
    auto read_frags = [&](auto bufc, auto ksc, bf16x8(&fw)[FN], bf16x8(&fa)[FM]) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, ks = decltype(ksc)::value;
        const uint16_t *bw = smem + BUF * SLOT, *ba = bw + BN * BK;
#pragma unroll
        for (int i = 0; i < FN; ++i)
            fw[i] = *reinterpret_cast<const bf16x8 *>(bw + (wn * FN + i) * MF * BK + (lane_off ^ (ks * CPK * 8)));
#pragma unroll
        for (int j = 0; j < FM; ++j)
            fa[j] = *reinterpret_cast<const bf16x8 *>(ba + (wm * FM + j) * MF * BK + (lane_off ^ (ks * CPK * 8)));
    };
    auto mfmas = [&](const bf16x8(&fw)[FN], const bf16x8(&fa)[FM]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                if constexpr ((ABL & 8) != 0) acc[i][j][0] += __builtin_bit_cast(float, (int)fw[i][0] ^ (int)fa[j][0]);
                else if constexpr (MF == 16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[i], fa[j], acc[i][j], 0, 0, 0);
            }
    };
    // K loop.  nk = (main iterations) * NST + L, L = NST + R (R = nk % NST, template parameter): the last L tiles are peeled so
    // that every slot index, wait count and "is there still a tile to request" is a compile-time constant and the steady-state
    // loop body is ONE basic block (run-time branches make hipcc drain lgkmcnt at every block boundary).
    constexpr int L = NST + R;
    const int n_main = (nk - L) / NST;
    // prologue: NST-1 tiles in flight (nk >= NST-1 required by the host)
    static_for<0, NST - 1>([&](auto bc) __attribute__((always_inline)) { stage(bc, decltype(bc)::value); });

    if constexpr (PIPE == 0) {
        auto tile = [&](auto bc, int kt, auto stagec, auto flyc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            if constexpr (UNEVEN) {
                if (full) WAIT_VM(decltype(flyc)::value * DPT);
                else WAIT_VM(decltype(flyc)::value * (DPT - 1));
            } else WAIT_VM(decltype(flyc)::value * DPT);
            __builtin_amdgcn_s_barrier();
            if constexpr (decltype(stagec)::value) stage(std::integral_constant<int, (b + NST - 1) % NST>{}, kt + NST - 1);
            constexpr int KSW = KS / WK;       // k-steps of this wave group
            bf16x8 fw[KSW][FN], fa[KSW][FM];
            if constexpr (WK == 1) {
                static_for<0, KS>([&](auto ksc) __attribute__((always_inline)) {
                    read_frags(bc, ksc, fw[decltype(ksc)::value], fa[decltype(ksc)::value]);
                });
            } else {
                const uint16_t *bw = smem + b * SLOT, *ba = bw + BN * BK;
#pragma unroll
                for (int t = 0; t < KSW; ++t) {
                    const int off = lane_off ^ ((wk * KSW + t) * CPK * 8);
#pragma unroll
                    for (int i = 0; i < FN; ++i) fw[t][i] = *reinterpret_cast<const bf16x8 *>(bw + (wn * FN + i) * MF * BK + off);
#pragma unroll
                    for (int j = 0; j < FM; ++j) fa[t][j] = *reinterpret_cast<const bf16x8 *>(ba + (wm * FM + j) * MF * BK + off);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) mfmas(fw[ks], fa[ks]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        int kt = 0;
        for (int it = 0; it < n_main; ++it) {
            static_for<0, NST>([&](auto bc) __attribute__((always_inline)) {
                tile(bc, kt + decltype(bc)::value, std::true_type{}, std::integral_constant<int, NST - 2>{});
            });
            kt += NST;
        }
        static_for<0, L>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr bool st = i <= R;                       // tile kt+i+NST-1 exists
            constexpr int fly = st ? NST - 2 : L - 1 - i;     // issued tiles younger than this one
            tile(std::integral_constant<int, i % NST>{}, kt + i, std::integral_constant<bool, st>{}, std::integral_constant<int, fly>{});
        });
    } else {
        static_assert(NST >= 3, "the k-step pipeline hands over to a tile requested one tile earlier");
        bf16x8 fw[2][FN], fa[2][FM];
        // PIPE 2: the fragment reads are inline asm and the waits are counted by hand (hipcc on gfx950 only ever emits
        // lgkmcnt(0), which would wait for the reads just issued for the NEXT k-step as well)
        const uint32_t lds0 = (uint32_t)(size_t)(const __attribute__((address_space(3))) uint16_t *)smem;
        uint32_t aw[KS][(NST + 1) / 2], aa[KS][(NST + 1) / 2];   // byte addresses: [k-step][slot pair]
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int h = 0; h < (NST + 1) / 2; ++h) {
                aw[ks][h] = lds0 + 2 * (h * 2 * SLOT + wn * FN * MF * BK + (lane_off ^ (ks * CPK * 8)));
                aa[ks][h] = lds0 + 2 * (h * 2 * SLOT + BN * BK + wm * FM * MF * BK + (lane_off ^ (ks * CPK * 8)));
            }
        auto read2 = [&](auto bufc, auto ksc, bf16x8(&w)[FN], bf16x8(&a)[FM]) __attribute__((always_inline)) {
            constexpr int BUF = decltype(bufc)::value, ks = decltype(ksc)::value;
            if constexpr (PIPE == 1) read_frags(bufc, ksc, w, a);
            else {
                static_assert(2 * ((BUF & 1) * SLOT + ((FN > FM ? FN : FM) - 1) * MF * BK) < 65536, "ds_read offset field");
                const uint32_t adw = aw[ks][BUF >> 1], ada = aa[ks][BUF >> 1];
#pragma unroll
                for (int i = 0; i < FN; ++i)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[i]) : "v"(adw), "n"(2 * ((BUF & 1) * SLOT + i * MF * BK)));
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[j]) : "v"(ada), "n"(2 * ((BUF & 1) * SLOT + j * MF * BK)));
            }
        };
        // all LDS reads except the NEWEST `newer` have returned; the fragments pass through the asm so that their users cannot
        // be scheduled above the wait
        auto wait2 = [&](auto newerc, bf16x8(&w)[FN], bf16x8(&a)[FM]) __attribute__((always_inline)) {
            if constexpr (PIPE == 2) {
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w[0]) : "n"(decltype(newerc)::value));
#pragma unroll
                for (int i = 1; i < FN; ++i) asm volatile("" : "+v"(w[i]));
#pragma unroll
                for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(a[j]));
            }
        };
        WAIT_VM((NST - 2) * DPT);
        __builtin_amdgcn_s_barrier();
        read2(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fw[0], fa[0]);
        // HO: 0 last tile (nothing follows), 1 hand over to the next tile without a request, 2 hand over and request tile kt+NST-1
        auto tile = [&](auto bc, int kt, auto hoc, auto flyc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value, HO = decltype(hoc)::value;
            static_for<0, KS>([&](auto ksc) __attribute__((always_inline)) {
                constexpr int ks = decltype(ksc)::value;
                constexpr bool more = ks + 1 < KS || HO > 0;
                if constexpr (ks + 1 < KS) {
                    read2(bc, std::integral_constant<int, ks + 1>{}, fw[(ks + 1) & 1], fa[(ks + 1) & 1]);
                } else if constexpr (HO > 0) {
                    // tile kt+1 must have landed for every wave; everyone has left tile kt-1, whose slot takes tile kt+NST-1
                    WAIT_VM(decltype(flyc)::value * DPT);
                    __builtin_amdgcn_s_barrier();
                    if constexpr (HO == 2) stage(std::integral_constant<int, (b + NST - 1) % NST>{}, kt + NST - 1);
                    read2(std::integral_constant<int, (b + 1) % NST>{}, std::integral_constant<int, 0>{}, fw[0], fa[0]);
                }
                wait2(std::integral_constant<int, more ? FN + FM : 0>{}, fw[ks & 1], fa[ks & 1]);
                mfmas(fw[ks & 1], fa[ks & 1]);
                if constexpr (PIPE == 1) {
                    // pin the issue order: the next k-step's fragment reads go out BEFORE this k-step's MFMAs
                    if constexpr (more) __builtin_amdgcn_sched_group_barrier(0x100, FN + FM, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, FN * FM, 0);
                }
            });
        };
        int kt = 0;
        for (int it = 0; it < n_main; ++it) {
            static_for<0, NST>([&](auto bc) __attribute__((always_inline)) {
                tile(bc, kt + decltype(bc)::value, std::integral_constant<int, 2>{}, std::integral_constant<int, NST - 3>{});
            });
            kt += NST;
        }
        static_for<0, L>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int HO = i == L - 1 ? 0 : (i <= R ? 2 : 1);
            constexpr int fly = HO == 2 ? NST - 3 : (L - i - 2 > 0 ? L - i - 2 : 0);   // issued tiles younger than kt+1
            tile(std::integral_constant<int, i % NST>{}, kt + i, std::integral_constant<int, HO>{}, std::integral_constant<int, fly>{});
        });
    }

    if constexpr (WK > 1) {   // the wave groups' partial accumulators meet in LDS (the ring is idle); group 0 carries on
        __syncthreads();
        f32x4 *red = reinterpret_cast<f32x4 *>(smem);
        static_assert(WK == 2, "one hand-over");
        if (wk == 1) {
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) red[((w2 * FN + i) * FM + j) * 64 + lane] = acc[i][j];
        }
        __syncthreads();
        if (wk == 1) return;
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                const f32x4 o = red[((w2 * FN + i) * FM + j) * 64 + lane];
                acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
            }
    }
    // epilogue: bias (+ GELU), 16-byte bf16 stores after a lane-pair exchange
    if (EPI >= 2) __syncthreads();
    const int nbase = n0 + wn * FN * MF, mbase = m0 + wm * FM * MF;
    if constexpr (MF == 16) {
        const int g = lane >> 4;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = mbase + j * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < FN / 2; ++q) {
                float w[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * q][j][r]), __float_as_uint(acc[2 * q + 1][j][r]), false, false);
                    w[r] = __uint_as_float(sw[0]);
                    w[4 + r] = __uint_as_float(sw[1]);
                }
                const int n = nbase + q * 32 + (g & 1) * 16 + (g >> 1) * 8;
                if (m >= M || n >= N || ((ABL & 1) && acc[0][0][0] != 12345.678f)) continue;
                const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + n), b1 = *reinterpret_cast<const float4 *>(p.bias + n + 4);
                w[0] += b0.x; w[1] += b0.y; w[2] += b0.z; w[3] += b0.w; w[4] += b1.x; w[5] += b1.y; w[6] += b1.z; w[7] += b1.w;
                if (EPI == 1 || EPI == 3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = gelu_erf(w[e]);
                }
                const uint4 pk = make_uint4(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]), pack_bf16x2(w[4], w[5]), pack_bf16x2(w[6], w[7]));
                if (EPI >= 2) *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(smem) + (size_t)(m - m0) * (BN * 2 + 32) + (n - n0) * 2) = pk;
                else *reinterpret_cast<uint4 *>(p.out + (size_t)m * p.ldo + n) = pk;
            }
        }
        if (EPI >= 2) {   // staged store: the tile leaves through LDS in whole 256-byte rows
            __syncthreads();
            constexpr int CPR = BN * 2 / 16;
            for (int idx = tid; idx < BM * CPR; idx += 64 * NW) {
                const int row = idx / CPR, c = idx % CPR;
                const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(smem) + (size_t)row * (BN * 2 + 32) + c * 16);
                if (m0 + row < M && n0 + c * 8 < N && !((ABL & 1) && acc[0][0][0] != 12345.678f))
                    *reinterpret_cast<uint4 *>(p.out + (size_t)(m0 + row) * p.ldo + n0 + c * 8) = v;
            }
        }
    } else {
        const int h = lane >> 5;
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int m = mbase + j * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {   // quads (2pq, 2pq+1) -> 8 consecutive columns from i*32 + (2pq+h)*8
                    float w[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][(2 * pq) * 4 + r]),
                                                                         __float_as_uint(acc[i][j][(2 * pq + 1) * 4 + r]), false, false);
                        w[r] = __uint_as_float(sw[0]);
                        w[4 + r] = __uint_as_float(sw[1]);
                    }
                    const int n = nbase + i * 32 + (2 * pq + h) * 8;
                    if (m >= M || n >= N || ((ABL & 1) && acc[0][0][0] != 12345.678f)) continue;
                    const float4 b0 = *reinterpret_cast<const float4 *>(p.bias + n), b1 = *reinterpret_cast<const float4 *>(p.bias + n + 4);
                    w[0] += b0.x; w[1] += b0.y; w[2] += b0.z; w[3] += b0.w; w[4] += b1.x; w[5] += b1.y; w[6] += b1.z; w[7] += b1.w;
                    if (EPI == 1) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] = gelu_erf(w[e]);
                    }
                    *reinterpret_cast<uint4 *>(p.out + (size_t)m * p.ldo + n) =
                        make_uint4(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]), pack_bf16x2(w[4], w[5]), pack_bf16x2(w[6], w[7]));
                }
        }
    }
}

// ---- reference + harness -------------------------------------------------------------------------------------------
__global__ void ref_kernel(P p, float *ref, int epi)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= p.N) return;
    float s = 0.f;
    for (int k = 0; k < p.K; ++k)
        s += __uint_as_float((uint32_t)p.A[(size_t)m * p.lda + k] << 16) * __uint_as_float((uint32_t)(p.wtiled ? p.W[(size_t)(n >> 3) * (p.K / 64) * 512 + (size_t)(k >> 6) * 512 + (n & 7) * 64 + (k & 63)] : p.W[(size_t)n * p.ldw + k]) << 16);
    s += p.bias[n];
    if (epi == 1 || epi == 3) s = 0.5f * s * (1.0f + erff(s * 0.70710678118654752f));
    ref[(size_t)m * p.N + n] = s;
}

__global__ void cmp_kernel(const uint16_t *out, const float *ref, size_t n, unsigned *bad, float *maxerr, int N, long long ldo)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o = __uint_as_float((uint32_t)out[(i / N) * ldo + (i % N)] << 16), r = ref[i];
    const float err = fabsf(o - r), tol = 0.02f + 0.012f * fabsf(r);
    if (!(err <= tol)) atomicAdd(bad, 1u);
    atomicMax(reinterpret_cast<unsigned *>(maxerr), __float_as_uint(err));
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

struct Shape { int M, N, K; const char *name; };
struct Bufs {
    uint16_t *A, *W, *out; float *bias, *ref; unsigned *bad; float *maxerr;
    int wcopies; size_t wstride;
};
static const char *g_filter = nullptr;
static int g_iters = 200, g_pada = 0, g_padw = 0, g_pado = 0, g_wtiled = 0, g_xmap = 0;

template <int MF, int WM, int WN, int FM, int FN, int NST, int PIPE, int EPI, int R, int ABL = 0, int WK = 1>
static void run_variant_r(const char *name, const Shape &s, Bufs &b)
{
    if (g_filter && !strstr(name, g_filter)) return;
    constexpr int BM = WM * FM * MF, BN = WN * FN * MF;
    constexpr size_t lds = (size_t)NST * (BM + BN) * 64 * 2;
    auto kern = lab_kernel<MF, WM, WN, FM, FN, NST, PIPE, EPI, R, ABL, WK>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    P p{s.M, s.N, s.K, b.A, b.W, b.bias, b.out, s.K + g_pada, s.N + g_pado, s.K + g_padw, g_wtiled, g_xmap};
    const dim3 grid((s.N + BN - 1) / BN, (s.M + BM - 1) / BM), block(64 * WM * WN * WK);
    // correctness on the full output
    CK(hipMemset(b.out, 0xff, (size_t)s.M * (s.N + g_pado) * 2));
    CK(hipMemset(b.bad, 0, 4)); CK(hipMemset(b.maxerr, 0, 4));
    hipLaunchKernelGGL(kern, grid, block, lds, 0, p);
    CK(hipGetLastError());
    ref_kernel<<<dim3((s.N + 255) / 256, s.M), 256>>>(p, b.ref, EPI);
    const size_t n = (size_t)s.M * s.N;
    cmp_kernel<<<(unsigned)((n + 255) / 256), 256>>>(b.out, b.ref, n, b.bad, b.maxerr, s.N, p.ldo);
    CK(hipDeviceSynchronize());
    unsigned bad; float maxerr;
    CK(hipMemcpy(&bad, b.bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&maxerr, b.maxerr, 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float us[2];
    for (int mode = 0; mode < 2; ++mode) {   // 0 cold weights (rotating copies), 1 warm
        const int copies = mode == 0 ? b.wcopies : 1;
        for (int it = 0; it < 10; ++it) { P q = p; q.W = b.W + (size_t)(it % copies) * b.wstride; hipLaunchKernelGGL(kern, grid, block, lds, 0, q); }
        CK(hipEventRecord(e0));
        for (int it = 0; it < g_iters; ++it) { P q = p; q.W = b.W + (size_t)(it % copies) * b.wstride; hipLaunchKernelGGL(kern, grid, block, lds, 0, q); }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        us[mode] = ms * 1e3f / g_iters;
    }
    const double fl = 2.0 * s.M * s.N * s.K;
    if (ABL) bad = 0;
    printf("pad a%d w%d o%d wt%d xm%d ", g_pada, g_padw, g_pado, g_wtiled, g_xmap);
    printf("%-34s %-18s grid %4d lds %6zu  cold %7.2f us %7.1f TF | warm %7.2f us %7.1f TF | %s maxerr %.3g\n", name, s.name,
           grid.x * grid.y, lds, us[0], fl / us[0] / 1e6, us[1], fl / us[1] / 1e6, bad ? "WRONG" : "ok", maxerr);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int MF, int WM, int WN, int FM, int FN, int NST, int PIPE, int EPI>
static void run_variant(const char *name, const Shape &s, Bufs &b)
{
    const int nk = s.K / 64, r = nk % NST;
    if (nk < 2 * NST) { return; }
    if (r == 0) run_variant_r<MF, WM, WN, FM, FN, NST, PIPE, EPI, 0>(name, s, b);
    else if (r == 1) { if constexpr (NST > 1) run_variant_r<MF, WM, WN, FM, FN, NST, PIPE, EPI, 1 % NST>(name, s, b); }
    else if (r == 2) { if constexpr (NST > 2) run_variant_r<MF, WM, WN, FM, FN, NST, PIPE, EPI, 2 % NST>(name, s, b); }
    else printf("%s: nk %% NST = %d not instantiated\n", name, r);
}

template <int MF, int WM, int WN, int FM, int FN, int NST, int PIPE, int ABL>
static void run_abl(const char *name, const Shape &s, Bufs &b)
{
    if ((s.K / 64) % NST) return;
    char nm[96];
    snprintf(nm, sizeof nm, "%s abl%d", name, ABL);
    run_variant_r<MF, WM, WN, FM, FN, NST, PIPE, 0, 0, ABL>(nm, s, b);
}
template <int MF, int WM, int WN, int FM, int FN, int NST, int PIPE>
static void run_abls(const char *name, const Shape &s, Bufs &b)
{
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 0>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 1>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 2>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 4>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 6>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 7>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 8>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 16>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 17>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 24>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 25>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 32>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 33>(name, s, b);
    run_abl<MF, WM, WN, FM, FN, NST, PIPE, 57>(name, s, b);
}

template <int MF, int WM, int WN, int FM, int FN, int NST>
static void run_wk2(const char *name, const Shape &s, Bufs &b)
{
    if ((s.K / 64) % NST || s.K / 64 < 2 * NST) return;
    run_variant_r<MF, WM, WN, FM, FN, NST, 0, 0, 0, 0, 2>(name, s, b);
}

int main(int argc, char **argv)
{
    if (argc > 1) g_filter = argv[1];
    if (argc > 2) g_iters = atoi(argv[2]);
    const Shape shapes[] = {{1536, 4096, 1024, "fc1 1536x4096x1024"}, {1536, 3072, 1024, "qkv 1536x3072x1024"},
                            {1536, 1024, 4096, "fc2 1536x1024x4096"}, {1536, 1024, 1024, "proj 1536x1024x1024"},
                            {768, 1024, 1024, "caq 768x1024x1024"},   {6144, 4096, 1024, "fc1x4 6144x4096x1024"}};
    if (getenv("LAB_PADA")) g_pada = atoi(getenv("LAB_PADA"));
    if (getenv("LAB_PADW")) g_padw = atoi(getenv("LAB_PADW"));
    if (getenv("LAB_PADO")) g_pado = atoi(getenv("LAB_PADO"));
    if (getenv("LAB_WTILED")) g_wtiled = atoi(getenv("LAB_WTILED"));
    if (getenv("LAB_XMAP")) g_xmap = atoi(getenv("LAB_XMAP"));
    const size_t maxA = (size_t)6144 * (4096 + 512), maxW = (size_t)4096 * 4096, maxO = (size_t)6144 * (4096 + 512);
    Bufs b;
    b.wcopies = 40;   // 40 x 8 MiB > the 256 MiB Infinity Cache
    b.wstride = (size_t)4096 * (1024 + 512);
    CK(hipMalloc(&b.A, maxA * 2)); CK(hipMalloc(&b.W, b.wstride * b.wcopies * 2)); CK(hipMalloc(&b.out, maxO * 2));
    CK(hipMalloc(&b.bias, 4096 * 4)); CK(hipMalloc(&b.ref, maxO * 4)); CK(hipMalloc(&b.bad, 4)); CK(hipMalloc(&b.maxerr, 4));
    {
        std::vector<uint16_t> h(maxA);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto &v : h) v = f2bf(rnd());
        CK(hipMemcpy(b.A, h.data(), maxA * 2, hipMemcpyHostToDevice));
        std::vector<uint16_t> w(b.wstride);
        for (auto &v : w) v = f2bf(rnd() * 0.05f);
        for (int c = 0; c < b.wcopies; ++c) CK(hipMemcpy(b.W + c * b.wstride, w.data(), b.wstride * 2, hipMemcpyHostToDevice));
        std::vector<float> bi(4096);
        for (auto &v : bi) v = rnd();
        CK(hipMemcpy(b.bias, bi.data(), 4096 * 4, hipMemcpyHostToDevice));
    }
    (void)maxW;
    for (const Shape &s : shapes) {
        if (getenv("LAB_SHAPE") && !strstr(s.name, getenv("LAB_SHAPE"))) continue;
        const int E = 0;
        if (getenv("LAB_WN1")) {
            run_variant<16, 2, 2, 3, 2, 4, 0, E>("m16  96x64 4w(48x32) nst4 batch", s, b);
            run_variant<16, 2, 1, 3, 4, 4, 0, E>("m16  96x64 2w(48x64) nst4 batch", s, b);
            run_variant<16, 2, 1, 3, 4, 4, 2, E>("m16  96x64 2w(48x64) nst4 asm", s, b);
            run_variant<16, 4, 1, 2, 4, 4, 0, E>("m16 128x64 4w(32x64) nst4 batch", s, b);
            run_variant<16, 4, 1, 2, 4, 4, 2, E>("m16 128x64 4w(32x64) nst4 asm", s, b);
            run_variant<16, 4, 1, 1, 4, 4, 0, E>("m16  64x64 4w(16x64) nst4 batch", s, b);
            run_variant<16, 4, 1, 1, 4, 4, 2, E>("m16  64x64 4w(16x64) nst4 asm", s, b);
            run_variant<16, 2, 1, 2, 4, 4, 0, E>("m16  64x64 2w(32x64) nst4 batch", s, b);
            run_variant<16, 2, 1, 2, 4, 4, 2, E>("m16  64x64 2w(32x64) nst4 asm", s, b);
            run_variant<16, 4, 1, 3, 4, 4, 2, E>("m16 192x64 4w(48x64) nst4 asm", s, b);
            run_variant<16, 2, 2, 2, 2, 4, 0, E>("m16  64x64 4w(32x32) nst4 batch", s, b);
            continue;
        }
        if (getenv("LAB_WK")) {   // round 4: the residual GEMMs' tile with a second wave group on the other k-step
            run_variant<16, 2, 2, 3, 2, 4, 0, E>("m16  96x64 4w nst4 batch", s, b);
            run_wk2<16, 2, 2, 3, 2, 4>("m16  96x64 8w(wk2, 3+2 DMA) nst4 batch", s, b);
            run_variant<16, 2, 2, 4, 2, 4, 0, E>("m16 128x64 4w nst4 batch", s, b);
            run_wk2<16, 2, 2, 4, 2, 4>("m16 128x64 8w(wk2) nst4 batch", s, b);
            run_variant<16, 2, 2, 2, 2, 4, 0, E>("m16  64x64 4w nst4 batch", s, b);
            run_wk2<16, 2, 2, 2, 2, 4>("m16  64x64 8w(wk2) nst4 batch", s, b);
            run_wk2<16, 2, 2, 2, 4, 4>("m16  64x128 8w(wk2) nst4 batch", s, b);
            run_wk2<16, 2, 2, 4, 4, 4>("m16 128x128 8w(wk2) nst4 batch", s, b);
            run_variant<16, 2, 2, 4, 4, 4, 0, E>("m16 128x128 4w nst4 batch", s, b);
            continue;
        }
        if (getenv("LAB_BEST")) {
            run_variant<16, 4, 2, 3, 4, 4, 2, E>("m16 192x128 8w(48x64) nst4 asm", s, b);
            run_variant<16, 2, 2, 3, 2, 4, 0, E>("m16  96x64 4w nst4 batch", s, b);
            run_variant<16, 2, 2, 3, 2, 4, 2, E>("m16  96x64 4w nst4 asm", s, b);
            run_variant<16, 2, 2, 2, 2, 4, 0, E>("m16  64x64 4w nst4 batch", s, b);
            run_variant<16, 2, 2, 3, 4, 2, 0, E>("m16  96x128 4w nst2 batch", s, b);
            continue;
        }
        if (getenv("LAB_DEEP")) {
            run_variant<16, 2, 2, 3, 2, 4, 0, E>("m16  96x64 4w nst4 batch", s, b);
            run_variant<16, 2, 2, 3, 2, 8, 0, E>("m16  96x64 4w nst8 batch", s, b);
            run_variant<16, 2, 2, 3, 2, 8, 2, E>("m16  96x64 4w nst8 asm", s, b);
            run_variant<16, 2, 2, 2, 2, 8, 0, E>("m16  64x64 4w nst8 batch", s, b);
            run_variant<16, 2, 2, 2, 2, 8, 2, E>("m16  64x64 4w nst8 asm", s, b);
            run_variant<16, 2, 2, 4, 2, 4, 0, E>("m16 128x64 4w nst4 batch", s, b);
            run_variant<16, 2, 2, 4, 2, 4, 2, E>("m16 128x64 4w nst4 asm", s, b);
            run_variant<16, 2, 2, 3, 4, 4, 0, E>("m16  96x128 4w nst4 batch", s, b);
            run_variant<16, 2, 2, 3, 4, 4, 2, E>("m16  96x128 4w nst4 asm", s, b);
            run_variant<16, 4, 2, 3, 2, 4, 2, E>("m16 192x64 8w nst4 asm", s, b);
            continue;
        }
        if (getenv("LAB_SMALL")) {
            run_variant<16, 2, 2, 3, 4, 2, 0, E>("m16  96x128 4w nst2 batch", s, b);
            run_variant<16, 2, 2, 3, 2, 4, 2, E>("m16  96x64 4w nst4 asm", s, b);
            run_variant<16, 2, 2, 3, 2, 4, 0, E>("m16  96x64 4w nst4 batch", s, b);
            run_variant<16, 2, 2, 3, 2, 2, 0, E>("m16  96x64 4w nst2 batch", s, b);
            run_variant<16, 2, 2, 2, 2, 4, 2, E>("m16  64x64 4w nst4 asm", s, b);
            run_variant<16, 2, 2, 2, 2, 4, 0, E>("m16  64x64 4w nst4 batch", s, b);
            run_variant<16, 2, 2, 2, 2, 2, 0, E>("m16  64x64 4w nst2 batch", s, b);
            run_variant<16, 2, 2, 2, 4, 4, 2, E>("m16  64x128 4w nst4 asm", s, b);
            run_variant<16, 2, 2, 2, 4, 4, 0, E>("m16  64x128 4w nst4 batch", s, b);
            run_variant<16, 2, 2, 2, 4, 2, 0, E>("m16  64x128 4w nst2 batch", s, b);
            run_variant<16, 4, 2, 1, 2, 4, 2, E>("m16  64x64 8w nst4 asm", s, b);
            run_variant<16, 4, 2, 2, 2, 4, 2, E>("m16 128x64 8w nst4 asm", s, b);
            run_variant<16, 4, 2, 1, 4, 4, 2, E>("m16  64x128 8w nst4 asm", s, b);
            run_variant<16, 1, 2, 3, 2, 4, 2, E>("m16  48x64 2w nst4 asm", s, b);
            run_variant<16, 1, 2, 3, 2, 4, 0, E>("m16  48x64 2w nst4 batch", s, b);
            run_variant<16, 2, 2, 1, 2, 4, 2, E>("m16  32x64 4w nst4 asm", s, b);
            continue;
        }
        if (getenv("LAB_FEW")) {
            run_variant<16, 4, 2, 3, 4, 4, 2, E>("m16 192x128 8w(48x64) nst4 asm", s, b);
            run_variant<16, 4, 2, 3, 4, 4, 2, 2>("m16 192x128 8w(48x64) nst4 asm STAGED", s, b);
            run_variant<16, 4, 2, 3, 4, 4, 2, 1>("m16 192x128 8w(48x64) nst4 asm GELU", s, b);
            run_variant<16, 4, 2, 3, 4, 4, 2, 3>("m16 192x128 8w(48x64) nst4 asm GELU STAGED", s, b);
            run_variant<16, 2, 2, 3, 4, 2, 0, 2>("m16  96x128 4w nst2 batch STAGED", s, b);
            run_variant<16, 2, 2, 3, 4, 2, 0, E>("m16  96x128 4w nst2 batch", s, b);
            run_variant<32, 2, 2, 3, 2, 4, 2, E>("m32 192x128 4w nst4 asm", s, b);
            run_variant<16, 2, 2, 4, 4, 2, 0, E>("m16 128x128 4w nst2 batch", s, b);
            run_variant<32, 4, 2, 2, 2, 3, 2, E>("m32 256x128 8w nst3 asm", s, b);
            continue;
        }
        if (getenv("LAB_ABL")) {
            run_abls<16, 4, 2, 3, 4, 4, 2>("m16 192x128 8w(48x64) nst4 asm", s, b);
            run_abls<16, 2, 2, 3, 4, 2, 0>("m16  96x128 4w nst2 batch", s, b);
            run_abls<32, 2, 2, 3, 2, 4, 2>("m32 192x128 4w nst4 asm", s, b);
            continue;
        }
        //            MF WM WN FM FN NST PIPE EPI
        run_variant<16, 2, 2, 4, 4, 4, 0, E>("m16 128x128 4w nst4 batch", s, b);
        run_variant<16, 2, 2, 4, 4, 2, 0, E>("m16 128x128 4w nst2 batch", s, b);
        run_variant<16, 2, 2, 3, 4, 2, 0, E>("m16  96x128 4w nst2 batch", s, b);
        run_variant<16, 2, 2, 6, 4, 4, 0, E>("m16 192x128 4w nst4 batch", s, b);
        run_variant<16, 2, 2, 6, 4, 3, 0, E>("m16 192x128 4w nst3 batch", s, b);
        run_variant<16, 2, 2, 6, 4, 4, 1, E>("m16 192x128 4w nst4 pipe", s, b);
        run_variant<16, 2, 2, 6, 4, 3, 1, E>("m16 192x128 4w nst3 pipe", s, b);
        run_variant<16, 2, 2, 4, 4, 4, 1, E>("m16 128x128 4w nst4 pipe", s, b);
        run_variant<32, 2, 2, 3, 2, 4, 0, E>("m32 192x128 4w nst4 batch", s, b);
        run_variant<32, 2, 2, 3, 2, 4, 1, E>("m32 192x128 4w nst4 pipe", s, b);
        run_variant<32, 2, 2, 3, 2, 4, 2, E>("m32 192x128 4w nst4 asm", s, b);
        run_variant<16, 2, 2, 6, 4, 4, 2, E>("m16 192x128 4w nst4 asm", s, b);
        run_variant<32, 2, 2, 2, 2, 4, 2, E>("m32 128x128 4w nst4 asm", s, b);
        run_variant<32, 4, 2, 2, 2, 3, 2, E>("m32 256x128 8w nst3 asm", s, b);
        run_variant<32, 2, 4, 3, 1, 4, 2, E>("m32 192x128 8w(96x32) nst4 asm", s, b);
        run_variant<16, 4, 2, 3, 4, 4, 2, E>("m16 192x128 8w(48x64) nst4 asm", s, b);
        run_variant<32, 2, 2, 4, 2, 3, 2, E>("m32 256x128 4w nst3 asm", s, b);
        run_variant<32, 2, 2, 3, 2, 3, 1, E>("m32 192x128 4w nst3 pipe", s, b);
        run_variant<32, 2, 2, 2, 2, 4, 1, E>("m32 128x128 4w nst4 pipe", s, b);
        run_variant<32, 2, 2, 2, 2, 2, 0, E>("m32 128x128 4w nst2 batch", s, b);
        run_variant<32, 2, 2, 4, 2, 3, 1, E>("m32 256x128 4w nst3 pipe", s, b);
        run_variant<32, 2, 2, 3, 3, 3, 1, E>("m32 192x192 4w nst3 pipe", s, b);
        run_variant<32, 2, 2, 4, 4, 2, 0, E>("m32 256x256 4w nst2 batch", s, b);
        run_variant<32, 4, 2, 2, 2, 3, 1, E>("m32 256x128 8w nst3 pipe", s, b);
        run_variant<32, 4, 2, 2, 2, 3, 0, E>("m32 256x128 8w nst3 batch", s, b);
        run_variant<32, 2, 4, 3, 1, 4, 1, E>("m32 192x128 8w(96x32) nst4 pipe", s, b);
        run_variant<32, 2, 4, 4, 2, 2, 0, E>("m32 256x256 8w nst2 batch", s, b);
        run_variant<16, 4, 2, 3, 4, 4, 1, E>("m16 192x128 8w(48x64) nst4 pipe", s, b);
        run_variant<16, 4, 2, 3, 4, 4, 0, E>("m16 192x128 8w(48x64) nst4 batch", s, b);
        run_variant<32, 2, 2, 3, 2, 4, 1, 1>("m32 192x128 4w nst4 pipe GELU", s, b);
        run_variant<16, 2, 2, 3, 4, 2, 0, 1>("m16  96x128 4w nst2 batch GELU", s, b);
    }
    return 0;
}
