// dit_attention.hip -- fused multi-head attention forward (no mask, no dropout), head_dim 64, bf16 MFMA, gfx950.
//
// Computes what the reference obtains from xformers.ops.memory_efficient_attention between the projections of
//   MemEffAttention.forward               /root/reference/vit/vision_transformer.py:284-297  (self-attention, L = 768)
//   MemoryEfficientCrossAttention.forward /root/reference/ldm/modules/attention.py:514-548   (image tokens, Lk = 1369)
// INCLUDING the per-head RMSNorm of q and k that precedes it (learned weight[64], eps 1e-5, dit/norm.py:29-43) when the
// caller has not already applied it in the projection GEMM, and the softmax scale 64^-1/2.
//
// MI355X mapping (flash-style, one pass over the keys, online softmax in fp32):
//   * grid (ceil(Lq / (16 NW)), heads, batch); NW query waves x KS key groups per workgroup, each wave owns 16 query rows
//     and every KS-th 64-key tile; a key group shares its staged K / V^T tiles in LDS;
//   * "swapped" products so that every reduction is lane-local or a lane swap: S^T = K Q^T puts one query in a lane
//     (column lane&15) with 4 keys per accumulator fragment, and O^T = V^T P^T keeps that query in the same lane, so
//     the running max / sum / rescale never cross lanes except for two v_permlane swaps per tile;
//   * P never leaves registers, and BOTH tiles are plain row-major images of global memory: the K fragment of MFMA row
//     i reads key row 32 (kf>>1) + 8 (i>>2) + 4 (kf&1) + (i&3) of the tile, which makes the 8 P values a lane holds for a
//     32-key block (fragments kf = 2 kb, 2 kb + 1) the 8 CONSECUTIVE keys 32 kb + 8 g .. + 7 -- exactly the 16-byte
//     V^T fragment (V arrives transposed from the projection GEMM's epilogue, keys contiguous);
//   * staging is LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip) into a ring of three
//     slots, two tiles ahead, with a counted vmcnt before the one barrier per tile (stamps of the register-staged
//     predecessor: ~400 of the ~2450 cycles of a step were the wait for loads issued one step earlier -- the six
//     workgroups of a (batch, head) sit on different XCDs, so every tile is an L2 miss -- and ~200 more the
//     VGPR -> LDS copy).  Rows are 128 bytes; the DMA writes LDS in lane order, so the bank-conflict-free image is made
//     on the SOURCE side: slot s of row r holds global 16-byte chunk s ^ ((r & 3) | ((r >> 3) & 1) << 2), conflict-free
//     for the hardware's ds_read_b128 lane groups under both row patterns (K: permuted, V^T: natural);
//   * a caller that wants K normalised here (k_norm_weight != NULL) gets the register-staged variant: K rows are
//     RMS-normalised by the 8 lanes that stage a row (3 xor-shuffles); Q rows by the 4 lane-groups that hold a row's
//     fragments; the softmax scale and log2(e) are folded into Q so the exponentials are bare v_exp_f32.
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "dit_common.h"

namespace gadit {

// GA_ATTN_STAMP builds record the cycle counter at eight points of one step (tools/attn_stamp.py); the product build has none.
#ifdef GA_ATTN_STAMP
__device__ unsigned long long g_attn_stamps[8 * 16 * 16];
#define STAMP(k)                                                                                                   \
    do {                                                                                                           \
        if (blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 8 && t_stamp == GA_ATTN_STAMP && lane == 0)              \
            g_attn_stamps[(blockIdx.x * 16 + wave) * 16 + (k)] = __builtin_readcyclecounter();                      \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

constexpr int KB = 64, HD = 64;
constexpr int TILE = KB * HD;  // elements of one staged tile (8 KiB)
#ifndef GA_ATTN_HEAD_MAJOR
#define GA_ATTN_HEAD_MAJOR 1
#endif
#ifndef GA_ATTN_ABLATE
#define GA_ATTN_ABLATE 0   // tools/attn_ablate.sh builds timing-only variants with phases removed (wrong results)
#endif
constexpr float kLazy = 8.f;    // log2 units a row maximum may exceed its reference before the reference moves

__device__ __forceinline__ int swz_of(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ swz_of(row)) * 8); }

__device__ __forceinline__ void glds16(const uint16_t *gsrc, uint16_t *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// max over the four lane groups that hold one query's keys (lanes l, l^16, l^32, l^48): two VALU lane swaps instead of two
// LDS round trips (ds_bpermute) on the tile's critical path
__device__ __forceinline__ float group_max(float t)
{
    const unsigned u = __float_as_uint(t);
    const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const unsigned v = __float_as_uint(m);
    const auto c = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}

// NW query waves x KS key groups per workgroup.  Wave (ks, wq) owns query rows wq*16 .. +15 of the workgroup's NW*16 and
// the 64-key tiles ks, ks + KS, ks + 2 KS, ...; every key group stages its own K / V^T tiles (KS rings in LDS) and the
// groups' partial (max, sum, O) are merged through LDS at the end.  KS > 1 only pays when 128-query workgroups would
// leave most CUs empty (see the launcher).  KNORM: K is RMS-normalised while it is staged (through registers).
// (tail: workgroups behind the attention grid -- y slices >= tail.y0, head-major; z slices >= tail.y0 otherwise -- compute a
//  ShiftBiasJob, dit_common.h; job.W[0] == nullptr: none)
struct AttnTail {
    ShiftBiasJob job;
    int y0;
    PrefetchJob pf;      // every tail workgroup takes its share (after its shift-bias part, if it has one)
    int nwgs;            // tail workgroups in all
    int pf_on;
};

template <int NW, int KS, bool KNORM>
__global__ __launch_bounds__(NW * KS * 64) void attention_fwd_kernel(GaAttentionArgs a, AttnTail tail)
{
    constexpr int QB = NW * 16, GT = NW * 64, CPT = (512 + GT - 1) / GT;  // 16-byte chunks per thread per staged tile
    constexpr int DPW = 16 / NW;                                          // DMA instructions per wave per tile (K + V^T)
    // tiles per barrier interval ("stage"): two when the ring still fits (3 slots x 2 tiles x 16 KiB = 96 KiB) -- in-situ
    // ablation put the per-tile barrier at ~20 % of the kernel
    constexpr int TPS = (!KNORM && KS == 1) ? 2 : 1;
    static_assert(16 % NW == 0, "a tile is 16 one-KiB DMA pieces");
    __shared__ __attribute__((aligned(16))) uint16_t smem[KS * 6 * TPS * TILE];  // per key group: K[3][TPS][key][d], V^T[3][TPS][d][key]
    constexpr int kTailPairs = sizeof(smem) / (kSbLdsFloats * sizeof(float)) >= 4 ? 4 : (int)(sizeof(smem) / (kSbLdsFloats * sizeof(float)));
    static_assert(kTailPairs >= 1, "the tail's partial sums");
    if (tail.job.W[0] != nullptr || tail.nwgs > 0) {   // kernel-uniform
#if GA_ATTN_HEAD_MAJOR
        const int slice = (int)blockIdx.y - tail.y0, in_slice = blockIdx.x, per_slice = gridDim.x;
#else
        const int slice = (int)blockIdx.z - tail.y0, in_slice = blockIdx.y * gridDim.x + blockIdx.x, per_slice = gridDim.x * gridDim.y;
#endif
        if (slice >= 0) {             // workgroup-uniform
            if (tail.job.W[0] != nullptr) shift_bias_block<kTailPairs>(tail.job, slice * per_slice + in_slice, reinterpret_cast<float *>(smem));
            if (tail.pf_on) prefetch_block(tail.pf, slice * per_slice + in_slice, tail.nwgs);
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: uniform branches
    const int ks = wave / NW, wq = wave - ks * NW, tg = tid - ks * GT;
    const int g = lane >> 4, c16 = lane & 15;
#if GA_ATTN_HEAD_MAJOR
    // grid (heads * batch, query tiles): the workgroups of one (batch, head) have ids heads * batch apart -- a multiple of 8 for
    // the DiT's 12 / 16 heads x even batch -- so they sit on ONE XCD and share its L2 copy of the head's K / V^T
    const int b = blockIdx.x / a.heads, h = blockIdx.x - b * a.heads, q0 = blockIdx.y * QB + wq * 16;
#else
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wq * 16;
#endif
    const int Lq = a.Lq, Lk = a.Lk;
    uint16_t *sK = smem + ks * 6 * TPS * TILE, *sV = sK + 3 * TPS * TILE;

    const int ntiles = (Lk + KB - 1) / KB, nfull = Lk / KB;
    const uint16_t *vt_base = a.vt + ((size_t)b * a.heads + h) * HD * a.vt_ld;
    const uint16_t *k_base = a.k + (size_t)b * Lk * a.k_stride + h * HD;
    int t_stamp = -1;
    (void)t_stamp;

    // ---- staging, DMA form: piece p (0..15) of a tile is 8 rows (p < 8: K rows 8p .., else V^T rows 8(p-8) ..); wave wq
    // of the key group moves pieces wq*DPW .. +DPW-1; lane l -> row 8p + (l>>3), LDS slot l&7 <- global chunk slot ^ swz
    // (round 6: the lane's byte offset inside a tile is computed ONCE -- 32 bits -- and a tile's address is a wave-uniform base plus that
    //  offset: the per-tile 64-bit multiply-adds of the first form were ~30 % of the VALU slots of a step, and the steps are VALU-bound)
    uint32_t dma_off[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int p = wq * DPW + i, r8 = (p & 7) * 8, row = r8 + (lane >> 3), chunk = (lane & 7) ^ swz_of(row);
        dma_off[i] = (uint32_t)row * (uint32_t)((p < 8 ? a.k_stride : a.vt_ld) * 2) + (uint32_t)chunk * 16u;
    }
    auto dma_tile = [&](int tile_raw, int slot) {
        const int tile = min(tile_raw, ntiles - 1);  // past the end: a harmless re-fetch (keeps the vmcnt arithmetic exact)
        const char *kt = reinterpret_cast<const char *>(k_base) + (size_t)tile * KB * (size_t)a.k_stride * 2;     // wave-uniform
        const char *vtt = reinterpret_cast<const char *>(vt_base) + (size_t)tile * KB * 2;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int p = wq * DPW + i, r8 = (p & 7) * 8;
            if (p < 8) {
                uint32_t off = dma_off[i];
                if (tile >= nfull) {      // wave-uniform: the ragged last tile re-reads the last key for the rows behind it (masked below)
                    const int row = r8 + (lane >> 3), chunk = (lane & 7) ^ swz_of(row);
                    off = (uint32_t)(min(tile * KB + row, Lk - 1) - tile * KB) * (uint32_t)(a.k_stride * 2) + (uint32_t)chunk * 16u;
                }
                glds16(reinterpret_cast<const uint16_t *>(kt + off), sK + slot * TILE + r8 * 64);
            } else {
                glds16(reinterpret_cast<const uint16_t *>(vtt + dma_off[i]), sV + slot * TILE + r8 * 64);
            }
        }
    };
    // a stage = TPS consecutive tiles in TPS consecutive tile buffers of ring slot `slot`
    auto dma_stage = [&](int stage, int slot) {
#pragma unroll
        for (int u = 0; u < TPS; ++u) dma_tile(stage * TPS + u, slot * TPS + u);
    };
    int s0 = 0, s1 = 1, s2 = 2;   // ring slots of the current stage, the next one, and the one being filled
    bool walk_primed = false;     // the projection below has already requested the key walk's first two stages

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + c16][kk*32 + g*8 .. +7], normalised, scaled
    bf16x8 qf[2];
    if (!KNORM && NW == 4 && TPS == 1 && a.qp_a != nullptr) {   // kernel-uniform
        // THE Q PROJECTION INSIDE THE WORKGROUP (round 5; GaAttentionArgs.qp_*): Q[64 queries x 64] of this head = A rows x W_head^T over
        // K = qp_k, instead of a GEMM launch of its own in front of the attention (the cross-attention of the denoiser: one launch and
        // the q round trip through memory less per block).  A 64-wide K-slice of W_head is 64 rows x 128 bytes -- the shape of a K tile
        // -- and so is the slice of the workgroup's 64 A rows: both are staged by the key walk's LDS-DMA pattern, W into the K ring of
        // the wave's key group (read with the K walk's permuted fragment rows), A into its V^T ring (read with the V^T walk's natural
        // rows), which leaves the accumulators of S^T = W_slice A_slice^T in exactly the lanes the Q fragments live in:
        // acc[kf][r] <-> Q[q0 + c16][32 (kf >> 1) + 8 g + 4 (kf & 1) + r].  The key groups split the K-slices (group ks takes slices
        // ks, ks + KS, ...) and their partial sums meet in LDS in a fixed order.  Ring: 3 slots, two slices ahead; one counted vmcnt
        // + one raw barrier per slice as in the key walk (everything is DMA: no compiler-managed load sits in the pipeline -- with the
        // A fragments as register loads hipcc put s_waitcnt vmcnt(0) in front of every DMA issue of the loop).  The request stream
        // simply continues into the key walk: request group G is W / A slice group G while G < steps_q and stage G - steps_q of the key
        // walk after that (both are four DMA instructions per wave), so the walk's first two K / V^T stages are already in flight
        // when the projection ends and it starts with the slots rotated to where they landed.
        constexpr int DPQ = 4;                      // DMA instructions per wave per slice: two W pieces + two A pieces of 8 rows
        const int nsl = a.qp_k >> 6, steps_q = (nsl + KS - 1) / KS;
        const int qbase = q0 - wq * 16;             // first query row of the workgroup
        float tot = 0.f;
        if (a.qp_row_ss) {   // RMSNorm row scale folded out of the A operand (same summation order as the GEMM consumer); used after the loop
            const float *rp = a.qp_row_ss + ((size_t)b * Lq + min(q0 + c16, Lq - 1)) * a.qp_row_ss_tiles;
            for (int t4 = 0; t4 < a.qp_row_ss_tiles; t4 += 4) {
                const float4 q4 = *reinterpret_cast<const float4 *>(rp + t4);
                tot += (q4.x + q4.y) + (q4.z + q4.w);
            }
        }
        static_assert(KNORM || TPS != 1 || NW != 4 || DPW * TPS == 4, "a stage of the key walk is four DMA instructions per wave, like a slice group");
        float qnw[16];       // per-head norm weights of my 16 head columns, requested in front of the DMA stream (a load behind it would wait for all of it)
        {
            float4 w4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w4[i] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.q_norm_weight) {   // kernel-uniform
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    w4[2 * kk] = *reinterpret_cast<const float4 *>(a.q_norm_weight + kk * 32 + g * 8);
                    w4[2 * kk + 1] = *reinterpret_cast<const float4 *>(a.q_norm_weight + kk * 32 + g * 8 + 4);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { qnw[4 * i] = w4[i].x; qnw[4 * i + 1] = w4[i].y; qnw[4 * i + 2] = w4[i].z; qnw[4 * i + 3] = w4[i].w; }
        }
        // (round 6: as in the key walk, a lane's byte offsets once, a slice's address = wave-uniform base + offset)
        uint32_t qw_off[2], qa_off[2];
        const char *qw_base[2];
        const char *qa_base = reinterpret_cast<const char *>(a.qp_a + (size_t)b * Lq * a.qp_lda);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wq * 2 + i, r8 = p * 8, row = r8 + (lane >> 3), chunk = (lane & 7) ^ swz_of(row);
            if (a.qp_w_tiled) {    // kernel-uniform
                qw_base[i] = reinterpret_cast<const char *>(a.qp_w + (size_t)(h * 8 + p) * nsl * 512);
                qw_off[i] = (uint32_t)((lane >> 3) * 64 + chunk * 8) * 2u;
            } else {
                qw_base[i] = reinterpret_cast<const char *>(a.qp_w + (size_t)h * 64 * a.qp_k);
                qw_off[i] = ((uint32_t)row * (uint32_t)a.qp_k + (uint32_t)chunk * 8u) * 2u;
            }
            qa_off[i] = ((uint32_t)min(qbase + row, Lq - 1) * (uint32_t)a.qp_lda + (uint32_t)chunk * 8u) * 2u;
        }
        const uint32_t qw_step = a.qp_w_tiled ? 1024u : 128u;     // bytes from one 64-wide K-slice of W to the next
        auto dma_q = [&](int grp, int slot) {
            if (grp >= steps_q) { dma_stage((grp - steps_q) * KS + ks, slot); return; }   // the key walk's stages 0, 1 (workgroup-uniform)
            const int sl = min(grp * KS + ks, nsl - 1);   // past the end: a harmless re-fetch (keeps the vmcnt arithmetic exact)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r8 = (wq * 2 + i) * 8;
                glds16(reinterpret_cast<const uint16_t *>(qw_base[i] + (size_t)sl * qw_step + qw_off[i]), sK + slot * TILE + r8 * 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r8 = (wq * 2 + i) * 8;
                glds16(reinterpret_cast<const uint16_t *>(qa_base + (size_t)sl * 128 + qa_off[i]), sV + slot * TILE + r8 * 64);
            }
        };
        f32x4 acc[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) acc[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int krow = 8 * (c16 >> 2) + (c16 & 3);
        dma_q(0, 0);
        dma_q(1, 1);
        for (int t = 0; t < steps_q; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPQ) : "memory");   // my pieces of slice group t; group t + 1 may be in flight
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (t * KS + ks < nsl) {   // group-uniform
                const uint16_t *bk = sK + s0 * TILE, *bx = sV + s0 * TILE;
                bf16x8 frag[4][2], xf[2];
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        frag[kf][kk] = *reinterpret_cast<const bf16x8 *>(bk + swz((kf >> 1) * 32 + (kf & 1) * 4 + krow, kk * 4 + g));
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) xf[kk] = *reinterpret_cast<const bf16x8 *>(bx + swz(wq * 16 + c16, kk * 4 + g));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int kf = 0; kf < 4; ++kf) acc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag[kf][kk], xf[kk], acc[kf], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            dma_q(t + 2, s2);     // the slot slice group t - 1 left (everybody is past this step's barrier)
            __builtin_amdgcn_sched_barrier(0);
            const int r = s0; s0 = s1; s1 = s2; s2 = r;
        }
        walk_primed = true;
        // s0 / s1 now name the slots the key walk's stages 0 / 1 are landing in; s2 held the last slice group
        if (KS > 1) {   // all key groups need the whole sum: partial accumulators through LDS, added in group order.  The exchange area is
                        // the FREE slot s2 of every group's rings (K tile: waves 0, 1; V^T tile: waves 2, 3) -- the others have DMA in flight
            __builtin_amdgcn_s_barrier();          // everybody has read the last slice group (raw: __syncthreads would drain vmcnt)
            auto xslot = [&](int k2) { return reinterpret_cast<float *>((wq < 2 ? smem + k2 * 6 * TPS * TILE : smem + k2 * 6 * TPS * TILE + 3 * TPS * TILE) + s2 * TILE) + (wq & 1) * 16 * 64 + lane; };
            float *mine = xslot(ks);
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(kf * 4 + r) * 64] = acc[kf][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float sum = 0.f;
#pragma unroll
                    for (int k2 = 0; k2 < KS; ++k2) sum += xslot(k2)[(kf * 4 + r) * 64];
                    acc[kf][r] = sum;
                }
            // (slot s2 is overwritten by the key walk's stage 2, requested behind the barrier of its first step: everybody has read by then)
        }
        const float rsc = a.qp_row_ss ? rsqrtf(tot * (1.0f / (float)a.qp_row_ss_dim) + a.qp_row_ss_eps) : 1.f;
        // row scale, per-head RMSNorm, bf16 as the projection GEMM would have stored it, then the softmax scale as below
        float xv[16];
        float ss = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = acc[2 * kk + (e >> 2)][e & 3] * rsc;
                xv[kk * 8 + e] = v;
                ss += v * v;
            }
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        const float rn = rsqrtf(ss * (1.0f / HD) + 1e-5f);
        const float rs = 0.125f * 1.4426950408889634f;  // 64^-1/2 * log2(e)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = xv[kk * 8 + e];
                if (a.q_norm_weight) v *= rn * qnw[kk * 8 + e];
                qf[kk][e] = (short)f32_to_bf16(bf16_to_f32(f32_to_bf16(v)) * rs);
            }
    } else {
        {
            const int row = min(q0 + c16, Lq - 1);
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
                    qf[kk][e] = (short)f32_to_bf16(v * rs);
                }
        }
    }

    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = 0.f, l_run = 0.f;  // the group's first tile sets the first reference maximum

    // ---- staging, register form (KNORM): 512 16-byte chunks per tile and operand; chunk c -> row c>>3, part c&7
    uint4 rk[CPT], rv[CPT];
    auto issue = [&](int tile_raw) {
        const int tile = min(tile_raw, ntiles - 1);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tg + GT * i, row = c >> 3, part = c & 7;
            const int key = min(tile * KB + row, Lk - 1);
            rk[i] = *reinterpret_cast<const uint4 *>(k_base + (size_t)key * a.k_stride + part * 8);
            rv[i] = *reinterpret_cast<const uint4 *>(vt_base + (size_t)row * a.vt_ld + tile * KB + part * 8);
        }
    };
    auto write_lds = [&](int slot) {
        uint16_t *dk = sK + slot * TILE, *dv = sV + slot * TILE;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tg + GT * i, row = c >> 3, part = c & 7;
            // K: RMS-normalise the row (8 consecutive lanes hold it)
            const uint32_t kw[4] = {rk[i].x, rk[i].y, rk[i].z, rk[i].w};
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
            const uint4 pk = make_uint4(pack_bf16x2(kv[0], kv[1]), pack_bf16x2(kv[2], kv[3]), pack_bf16x2(kv[4], kv[5]),
                                        pack_bf16x2(kv[6], kv[7]));
            *reinterpret_cast<uint4 *>(dk + swz(row, part)) = pk;
            *reinterpret_cast<uint4 *>(dv + swz(row, part)) = rv[i];
        }
    };

    // One 64-key tile.  TAIL = the tile holds keys >= Lk (masked to -inf); full tiles carry no masking code at all.
    // Softmax bookkeeping is "lazy": the S^T accumulators start at -m_run (the C operand of the first MFMA, so the
    // subtraction is free), and as long as no row's tile maximum exceeds its reference m_run by more than kLazy (2^8 --
    // harmless in fp32 sums and in the relative precision of the bf16 P) the reference is kept: no max update, no
    // exp2(m_old - m_new), no rescale of the 16 O accumulators.  m_run is always the true maximum at the time it was
    // set, so the largest P of a row lies in [1, 256] and the row sum is >= 1.  The decision is wave-uniform.
    auto tile_math = [&](int tile, bool first, int slot, int dma_stage_idx, int dma_slot, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        const uint16_t *bk = sK + slot * TILE, *bv = sV + slot * TILE;
        // The fragment reads are issued in two batches into their own registers (the compiler, left alone, funnels them
        // through one register quad: read -> wait -> MFMA, eight LDS latencies in a row, twice per tile); the V^T
        // fragments are requested as soon as the S MFMAs have consumed the K fragments and land while the softmax runs.
        bf16x8 frag[4][2];
        const int krow = 8 * (c16 >> 2) + (c16 & 3);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                frag[kf][kk] = *reinterpret_cast<const bf16x8 *>(bk + swz((kf >> 1) * 32 + (kf & 1) * 4 + krow, kk * 4 + g));
        __builtin_amdgcn_sched_barrier(0);
        STAMP(2);

        // ---- S^T - m_run = K Q^T - m_run : s[kf][r] <-> key (kf>>1)*32 + g*8 + (kf&1)*4 + r of the tile, query c16
        f32x4 s[4];
        const float nm = -m_run;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) s[kf] = f32x4{nm, nm, nm, nm};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
                if (GA_ATTN_ABLATE != 2) s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag[kf][kk], qf[kk], s[kf], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        // the next-but-one tile's DMA is requested here, behind the head of the step's dependency chain (K fragments ->
        // S MFMAs) rather than in front of it
        if (!KNORM && GA_ATTN_ABLATE != 5 && dma_stage_idx >= 0) dma_stage(dma_stage_idx, dma_slot);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int df = 0; df < 4; ++df)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                if (GA_ATTN_ABLATE != 4) frag[df][kb] = *reinterpret_cast<const bf16x8 *>(bv + swz(df * 16 + c16, kb * 4 + g));
        __builtin_amdgcn_sched_barrier(0);

        if (TAIL) {
            const int kbase = tile * KB + g * 8;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kbase + (kf >> 1) * 32 + (kf & 1) * 4 + r >= Lk) s[kf][r] = -1e30f;
        }
        float tmax = fmaxf(s[0][0], s[0][1]);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[kf][r]);
        tmax = group_max(tmax);
        STAMP(4);
        if (__builtin_amdgcn_ballot_w64(first || tmax > kLazy) != 0) {  // rare after the first tile
            const float delta = first ? tmax : fmaxf(tmax, 0.f);
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
            m_run += delta;
            l_run *= alpha;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                s[kf][0] -= delta; s[kf][1] -= delta; s[kf][2] -= delta; s[kf][3] -= delta;
            }
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                o[df][0] *= alpha; o[df][1] *= alpha; o[df][2] *= alpha; o[df][3] *= alpha;
            }
        }
        bf16x8 pf[2];
        f32x2_t psum = {0.f, 0.f};
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2_t p = GA_ATTN_ABLATE == 3 ? f32x2_t{s[kf][r], s[kf][r + 1]}
                                                      : f32x2_t{__builtin_amdgcn_exp2f(s[kf][r]), __builtin_amdgcn_exp2f(s[kf][r + 1])};
                psum += p;
                pf[kf >> 1][(kf & 1) * 4 + r] = (short)f32_to_bf16(p[0]);
                pf[kf >> 1][(kf & 1) * 4 + r + 1] = (short)f32_to_bf16(p[1]);
            }
        l_run += psum[0] + psum[1];
        // ---- O^T += V^T P^T : o[df][r] = O[q = c16][d = df*16 + g*4 + r]; the lane's 8 P of block kb are keys 32 kb + 8 g ..
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int df = 0; df < 4; ++df)
                if (GA_ATTN_ABLATE != 1) o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag[df][kb], pf[kb], o[df], 0, 0, 0);
                else o[df][0] += __builtin_bit_cast(float, (int)frag[df][kb][0] ^ (int)pf[kb][0]);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(6);
    };

    // ---- the walk: step t works on stage t*KS + ks (TPS tiles) in ring slot t % 3 while stage t+2 (DMA) / tile t+1 (registers)
    // is staged.  DMA form: counted wait for this wave's pieces of stage t, then a RAW s_barrier (everyone's pieces have
    // landed and everyone has left stage t-1, whose slot the next DMA overwrites) -- __syncthreads() would drain vmcnt to 0.
    if (KNORM) {
        issue(ks);
        write_lds(0);
        issue(KS + ks);
    } else if (!walk_primed) {
        dma_stage(ks, 0);
        dma_stage(KS + ks, 1);
    }
    const int nstages = (ntiles + TPS - 1) / TPS, steps = (nstages + KS - 1) / KS;
    for (int t = 0; t < steps; ++t) {
        const int stage = t * KS + ks;
        t_stamp = t;
        STAMP(0);
        if (KNORM) {
            __syncthreads();
            write_lds(s1);                        // tile t+1 into the slot tile t-2 left two barriers ago
            issue(stage + 2 * KS);
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GA_ATTN_ABLATE == 5 ? 0 : DPW * TPS) : "memory");
            if (GA_ATTN_ABLATE != 6) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(1);
        bool dma_done = false;
#pragma unroll
        for (int u = 0; u < TPS; ++u) {
            const int tile = stage * TPS + u;
            const int dma_idx = dma_done ? -1 : stage + 2 * KS;   // the stage after next goes out behind the first S MFMAs
            if (tile < nfull) { tile_math(tile, t == 0 && u == 0, s0 * TPS + u, dma_idx, s2, std::false_type{}); dma_done = true; }
            else if (tile < ntiles) { tile_math(tile, t == 0 && u == 0, s0 * TPS + u, dma_idx, s2, std::true_type{}); dma_done = true; }
        }
        if (!KNORM && GA_ATTN_ABLATE != 5 && !dma_done) dma_stage(stage + 2 * KS, s2);   // keeps the vmcnt arithmetic exact
        STAMP(7);
        const int r = s0; s0 = s1; s1 = s2; s2 = r;
    }
    if (!KNORM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may outlive the workgroup's LDS allocation
    if (KS > 1) __syncthreads();                                   // ... nor land in the exchange area below

    // ---- merge the key groups' partial softmax states (group 0 ends up with the total); the LDS tiles are dead
    if (KS > 1) {
        float *xch = reinterpret_cast<float *>(smem);  // [KS-1][NW][18][64]
        if (ks > 0) {
            float *dst = xch + ((size_t)(ks - 1) * NW + wq) * 18 * 64 + lane;
            dst[0] = ks < ntiles ? m_run : -1e30f;
            dst[64] = l_run;
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(2 + df * 4 + r) * 64] = o[df][r];
        }
        __syncthreads();
        if (ks > 0) return;
        float m_all = m_run;  // group 0 always has a tile (Lk >= 1)
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2) m_all = fmaxf(m_all, xch[((size_t)(k2 - 1) * NW + wq) * 18 * 64 + lane]);
        const float w0 = __builtin_amdgcn_exp2f(m_run - m_all);
        l_run *= w0;
#pragma unroll
        for (int df = 0; df < 4; ++df) { o[df][0] *= w0; o[df][1] *= w0; o[df][2] *= w0; o[df][3] *= w0; }
#pragma unroll
        for (int k2 = 1; k2 < KS; ++k2) {
            const float *src = xch + ((size_t)(k2 - 1) * NW + wq) * 18 * 64 + lane;
            const float wk = __builtin_amdgcn_exp2f(src[0] - m_all);
            l_run += wk * src[64];
#pragma unroll
            for (int df = 0; df < 4; ++df)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[df][r] += wk * src[(2 + df * 4 + r) * 64];
        }
    }

    float l = l_run;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int row = q0 + c16;
    if (row < Lq) {
        uint16_t *op = a.out + ((size_t)b * Lq + row) * a.out_stride + h * HD + g * 4;
#pragma unroll
        for (int df = 0; df < 4; ++df) {
            const uint2 p = make_uint2(pack_bf16x2(o[df][0] * inv, o[df][1] * inv), pack_bf16x2(o[df][2] * inv, o[df][3] * inv));
            *reinterpret_cast<uint2 *>(op + df * 16) = p;
        }
    }
}

template <int NW, int KS>
static void launch_attention(const GaAttentionArgs &a, hipStream_t s, const ShiftBiasJob *job, const PrefetchJob *pf = nullptr, int pf_wgs = 0)
{
    // K normalised while it is staged (not on the DiT path, which normalises K once per conditioning tensor): ONE instantiation, eight
    // query waves and a single key group -- <8,2,true> / <4,3,true> spilt registers and <4,1,true> carried a private segment (round-3 review)
    constexpr int QW = 8;
    const bool knorm = a.k_norm_weight != nullptr;
    const int rows = (knorm ? QW : NW) * 16;
#if GA_ATTN_HEAD_MAJOR
    dim3 grid(a.heads * a.batch, (a.Lq + rows - 1) / rows, 1);
#else
    dim3 grid((a.Lq + rows - 1) / rows, a.heads, a.batch);
#endif
    AttnTail tail{};
    if (pf && pf_wgs > 0) { tail.pf = *pf; tail.pf_on = 1; } else pf_wgs = 0;
    if (job || pf_wgs) {
        if (job) tail.job = *job;
        const int needed = std::max(job ? shift_bias_wgs(job->N0, job->N1) : 0, pf_wgs);
#if GA_ATTN_HEAD_MAJOR
        const int per_slice = (int)grid.x;
        tail.y0 = grid.y; grid.y += (needed + per_slice - 1) / per_slice;
        tail.nwgs = (int)(grid.y - tail.y0) * per_slice;
#else
        const int per_slice = (int)(grid.x * grid.y);
        tail.y0 = grid.z; grid.z += (needed + per_slice - 1) / per_slice;
        tail.nwgs = (int)(grid.z - tail.y0) * per_slice;
#endif
    }
    if (knorm) hipLaunchKernelGGL((attention_fwd_kernel<QW, 1, true>), grid, dim3(QW * 64), 0, s, a, tail);
    else hipLaunchKernelGGL((attention_fwd_kernel<NW, KS, false>), grid, dim3(NW * KS * 64), 0, s, a, tail);
}

}  // namespace gadit

#ifdef GA_ATTN_STAMP
extern "C" int ga_attn_debug_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gadit::g_attn_stamps), sizeof(unsigned long long) * 8 * 16 * 16);
}
#endif

namespace gadit {
static int dispatch_attention(const GaAttentionArgs *a, const ShiftBiasJob *job, void *stream, const PrefetchJob *pf = nullptr, int pf_wgs = 0);
// workgroups of the launch dispatch_attention picks for this shape (no k-norm): a tail only pays while they leave CUs idle
int attention_workgroups(const GaAttentionArgs *a)
{
    const int64_t wgs128 = (int64_t)((a->Lq + 127) / 128) * a->heads * a->batch;
    return (int)(wgs128 <= 128 ? (int64_t)((a->Lq + 63) / 64) * a->heads * a->batch : wgs128);
}
// the q projection can ride inside the attention workgroups (GaAttentionArgs.qp_*) when the launch takes the 64-query configuration
bool attention_fuses_q(const GaAttentionArgs *a)
{
    return (int64_t)((a->Lq + 127) / 128) * a->heads * a->batch <= 128;
}
int attention_with_tail(const GaAttentionArgs *a, const ShiftBiasJob *job, void *stream, const PrefetchJob *pf, int pf_wgs)
{
    if (job) {
        if (!job->W[0] || !job->W[1] || !job->shift || !job->out) return GA_DIT_ERR_NULL_ARG;
        if (job->N0 % 8 != 0 || job->N1 % 8 != 0 || job->K % 64 != 0 || job->K > 2048 || job->B <= 0) return GA_DIT_ERR_BAD_SHAPE;   // (K / 64 <= 4 x 8 waves)
    }
    if (pf)
        for (int r = 0; r < kPfRanges; ++r)
            if (pf->ptr[r] && pf->bytes[r] % 1024) return GA_DIT_ERR_BAD_SHAPE;
    return dispatch_attention(a, job, stream, pf, pf_wgs);
}
}  // namespace gadit

extern "C" int ga_attention_bf16(const GaAttentionArgs *a, void *stream) { return gadit::dispatch_attention(a, nullptr, stream); }

static int gadit::dispatch_attention(const GaAttentionArgs *a, const ShiftBiasJob *job, void *stream, const PrefetchJob *pf, int pf_wgs)
{
    using namespace gadit;
    if (!a || (!a->q && !a->qp_a) || !a->k || !a->vt || !a->out) return GA_DIT_ERR_NULL_ARG;
    if (a->qp_a) {   // the q projection inside the workgroup: LDS-DMA path only, 16-byte aligned operands, K in 64-wide slices
        if (!a->qp_w || a->k_norm_weight) return GA_DIT_ERR_NULL_ARG;
        if (a->qp_k < 64 || a->qp_k % 64 || a->qp_lda % 8 || a->qp_lda < a->qp_k || ((uintptr_t)a->qp_a | (uintptr_t)a->qp_w) % 16 != 0 ||
            (a->qp_row_ss && (a->qp_row_ss_tiles <= 0 || a->qp_row_ss_tiles % 4 != 0 || a->qp_row_ss_dim <= 0 || (uintptr_t)a->qp_row_ss % 16 != 0)))
            return GA_DIT_ERR_BAD_SHAPE;
    }
    if (a->batch <= 0 || a->heads <= 0 || a->Lq <= 0 || a->Lk <= 0 || a->q_stride % 8 || a->k_stride % 8 || a->vt_ld % 8 ||
        a->vt_ld < ((a->Lk + KB - 1) / KB) * KB || a->out_stride % 4)
        return GA_DIT_ERR_BAD_SHAPE;
    // 16-byte accesses (vector loads of q, LDS-DMA of k / vt, 8-byte stores of out)
    if (((a->qp_a ? 0 : (uintptr_t)a->q) | (uintptr_t)a->k | (uintptr_t)a->vt) % 16 != 0 || (uintptr_t)a->out % 8 != 0) return GA_DIT_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // 128-query workgroups when they fill the chip; otherwise (the conditional half of a CFG pair alone in the image
    // cross-attention: 6 x 16 x 1 = 96 workgroups for 256 CUs) 64-query workgroups with two key groups -- measured on
    // MI355X, B = 1, 16 heads, 768 x 1369: 20.7 -> 15.0 us; B = 2: 22.3 us (<8,1>) vs 28.5 us (<4,2>).
#ifdef GA_TUNING  // tuning builds only: NW*10 + KS from the environment
    const int cfg = [] { const char *e = getenv("GA_ATTN_CFG"); return e ? atoi(e) : 0; }();
#else
    constexpr int cfg = 0;
#endif
    const int64_t wgs128 = (int64_t)((a->Lq + 127) / 128) * a->heads * a->batch;
    if (a->qp_a && (cfg != 0 || wgs128 > 128)) return GA_DIT_ERR_BAD_SHAPE;   // only the 64-query configuration projects q itself (attention_fuses_q)
#ifdef GA_TUNING
    if (cfg == 23) { launch_attention<2, 3>(*a, s, job, pf, pf_wgs); return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH; }
    if (cfg == 43) { launch_attention<4, 3>(*a, s, job, pf, pf_wgs); return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH; }
    if (cfg == 22) { launch_attention<2, 2>(*a, s, job, pf, pf_wgs); return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH; }
    if (cfg == 41) { launch_attention<4, 1>(*a, s, job, pf, pf_wgs); return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH; }
    if (cfg == 82) { launch_attention<8, 2>(*a, s, job, pf, pf_wgs); return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH; }
    if (cfg == 21) { launch_attention<2, 1>(*a, s, job, pf, pf_wgs); return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH; }
#endif
    // Round 3 (tools/attn_cfg_sweep.py): three key groups on the small grids (1 x 16 x 768 x 1369: 15.9 -> 14.2 us, x 768 keys
    // 10.1 -> 9.6 us), and two key groups beside eight query waves while 128-query workgroups are fewer than two per CU
    // (2 x 16 x 768 x 768: 13.7 -> 12.8 us; 4 x 16 x 768 x 1369: 39.5 -> 37.1 us); one group on the grids beyond that.
    if (cfg == 42) launch_attention<4, 2>(*a, s, job, pf, pf_wgs);
    else if (cfg == 81) launch_attention<8, 1>(*a, s, job, pf, pf_wgs);
    else if (wgs128 <= 128) launch_attention<4, 3>(*a, s, job, pf, pf_wgs);
    else if (wgs128 <= 512) launch_attention<8, 2>(*a, s, job, pf, pf_wgs);
    else launch_attention<8, 1>(*a, s, job, pf, pf_wgs);
    return hipGetLastError() == hipSuccess ? GA_DIT_OK : GA_DIT_ERR_LAUNCH;
}
