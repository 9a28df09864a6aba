// surfel_common.h -- internal definitions shared by the surfel rasterizer translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ga_surfel.h"

namespace ga {

constexpr int kTile = 16;              // tile edge in pixels: fixes which pixels a splat may touch (upstream BLOCK_X/Y)
constexpr int kRec = GA_SURFEL_RECORD_FLOATS;
// blend-ready record, 96 bytes (6 x float4), 16-byte aligned.  The ray/splat intersection of upstream's blend,
//   k = px*Tw - Tu ; l = py*Tw - Tv ; p = cross(k, l),   is bilinear in the pixel:  p = (px-ox)*A + (py-oy)*B + C
//   with (ox,oy) = rint(centre), U = Tu - ox*Tw, V = Tv - oy*Tw, A = V x Tw, B = Tw x U, C = U x V  (pixel-independent,
//   so computed once per splat by the preprocess kernel):
//   q0 = A.x A.y B.x B.y | q1 = C.x C.y A.z B.z | q2 = cx cy C.z opacity      <- needed for every evaluated pair
//   q3 = Tw.x Tw.y Tw.z cull | q4 = n.x n.y n.z r | q5 = g b - -                 <- Tw, cull: every pair; rest: contributing pairs
// Values that are combined by one packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 take even-aligned register
// pairs) sit in the same half of a quad: (A.x,A.y) (B.x,B.y) (C.x,C.y) (cx,cy) (Tw.x,Tw.y) (n.x,n.y) (n.z,r) (g,b).
// `cull` = two fp16 half-extents (rx, ry), rounded up, of the conservative {alpha >= 1/255} pixel box about (cx, cy):
// the blend loop rejects pixel columns / rows with |x - cx| > rx or |y - cy| > ry without evaluating the pair
// (+inf = no bound, negative = the splat can never pass the threshold).
constexpr float kNear = 0.2f;          // upstream near_n
constexpr float kFar = 100.0f;         // upstream far_n
constexpr float kCutoff = 3.0f;
constexpr float kFilterSize = 0.707106f;
constexpr float kFilterInvSquare = 2.0f;

#ifndef GA_BIN_SPLATS
#define GA_BIN_SPLATS 8
#endif
constexpr int kBinSplats = GA_BIN_SPLATS;           // splats per thread in the fill kernel (2048 per workgroup)
#ifndef GA_PRE_SPLATS
#define GA_PRE_SPLATS 2
#endif
constexpr int kPreSplats = GA_PRE_SPLATS;  // groups of 256 Gaussians per workgroup of the preprocess kernel (1, 2 or 4)
#ifndef GA_PRE_VIEWS
#define GA_PRE_VIEWS 2
#endif
constexpr int kPreViews = GA_PRE_VIEWS;    // views a thread of the preprocess kernel walks with its Gaussian's camera-independent part in registers
constexpr int kViewSlots = 64;          // words the per-view entry count is spread over (same-address atomics serialise)
constexpr int kLdsTiles = 8192;         // per-view tile counters aggregated in LDS up to this many tiles (32 KiB)
// Segmented blend: lists of >= kLongList entries (length class >= kSegClass; class b holds 2^(b-1) <= n < 2^b) are cut into
// seg_count(b) segments of 256..512 entries, each blended by its own workgroup; a segment (<= kItemChunks chunks of 64)
// stays resident in LDS for both of its passes, a shorter unsegmented list streams through the same LDS as a ring.
#ifndef GA_ITEM_CHUNKS
#define GA_ITEM_CHUNKS 8
#endif
constexpr int kItemChunks = GA_ITEM_CHUNKS;   // 8: three blend workgroups per CU (47 KB of LDS each); 6: four (35 KB) -- segments then hold <= 384 entries
#ifndef GA_SEG_CLASS
#define GA_SEG_CLASS 12
#endif
constexpr int kSegClass = GA_SEG_CLASS;
constexpr int kLongList = 1 << (GA_SEG_CLASS - 1);
// class b holds 2^(b-1) <= n < 2^b entries: 2^(b-9) segments of 256 .. 512 entries (8 chunks), or 3 * 2^(b-10) of 171 .. 341 (6 chunks)
__host__ __device__ constexpr uint32_t seg_count(int b) { return b < kSegClass ? 1u : (kItemChunks >= 8 ? (1u << (b - 9)) : (3u << (b - 10))); }
// launch epoch of the cross-workgroup exchange words: a DEVICE word (seg_table[kSegEpochWord], behind the 2 x 40 table entries) that
// the tile scan bumps once per launch and the per-launch memset does not touch -- a host-side counter passed as a kernel
// argument is frozen under HIP-graph replay, and words of the previous replay would validate
// (tile, segment) work items the exchange scratch holds: the caller's seg_capacity, or by default 1/8 of the worst case
// capacity / 256 (every entry in a list of 2048 or more) plus a floor; the tile scan reports more as an overflow
__host__ __device__ constexpr int64_t seg_items(int64_t capacity, int64_t seg_capacity)
{
    return seg_capacity > 0 ? seg_capacity : (kItemChunks >= 8 ? capacity / 2048 : capacity / 1365) + 128;
}
constexpr int kSegEpochWord = 96;
constexpr int kSegTableWords = 128;
constexpr int kSegFloats = 15 * 256;     // scratch words per segment: transmittance + 14 partial sums for 256 pixels
constexpr int kSortCap = GA_SURFEL_SORT_RUN;         // per-tile entries sorted in one LDS pass (16 KiB of u64 keys: 8+ workgroups per CU)

struct Dims {
    int N, V, H, W, gx, gy, tiles;     // tiles = gx*gy per view
};

struct Workspace {
    int64_t *status;
    uint32_t *seg_sync;   // [8 * ntile_cap] per segmented tile: arrival counters of the four quadrants, saturation word
    uint32_t *seg_table;  // [2 * 40] per class b: (first tile_order slot, first segment work item)
    unsigned long long *seg_scratch;   // (value, launch epoch) words, see surfel_blend.hip
    uint32_t *tile_count, *tile_start, *tile_cursor;
    unsigned long long *view_total;   // [V][kViewSlots] entries per view (sum of its tile counters) in kViewSlots partial counts
    uint4 *tile_order;   // schedule of the per-tile kernels, longest lists first: (tile, list begin, list length, 0)
    uint4 *run_table;    // runs 1.. of the lists longer than one sort run: (tile, run, list begin, list length)
    uint16_t *rect;
    float *depth, *record;
    uint64_t *keys;
    uint32_t *point_list;
};

// Bijective XCD-aware remap: consecutive workgroup ids land on different XCDs (id % 8); give each XCD one contiguous
// span of logical ids so that neighbouring tiles (which share splat records) share that XCD's L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg)
{
    constexpr uint32_t X = 8;
    const uint32_t q = nwg / X, r = nwg % X, xcd = bid % X, k = bid / X;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

void launch_preprocess(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s);
void launch_binning(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s);
void launch_tile_sort(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s);
void launch_blend(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s);

}  // namespace ga
