// surfel_backward.hip -- backward pass of the 2D-surfel rasterizer, gfx950.
//
// What the reference's training call sites differentiate through: GaussianRasterizer(...)(means3D, means2D, ...) at
// /root/reference/nsr/gs_surfel.py:104-114 (upstream diff_surfel_rasterization backward.cu, third-party and absent).  The
// forward being differentiated is SURVEY.md Appendix A.1; the piecewise-constant choices are treated as constants of the
// gradient exactly as oracle/surfel_autograd.py (the backward oracle) does: the selection min(rho3d, rho2d), the
// alpha >= 1/255, depth >= near and T(1 - alpha) >= 1e-4 tests, the 0.99 clamp (no gradient above it), the sign that turns
// the normal towards the camera, and WHICH pair is the median contributor (its depth carries the gradient of the median-depth
// channel, as in upstream's backward).
//
// Launches (all on the caller's stream, nothing allocated):
//   1. surfel_bwd_record_kernel      per (view, Gaussian): the forward's per-splat quantities once more -- Tu, Tv, Tw, the
//      screen-space centre, the camera-facing normal, opacity, colour, the forward's cull half-extents -- as a 24-float record;
//   2. (an extra grid row of the same launch) every tile's depth-ordered list cut into 128-entry segments: first segment of
//      each tile, owner of each segment;
//   3. the blend backward, ONE WORKGROUP PER SEGMENT in each of three launches (see the comment above BwdShared) with a
//      one-workgroup-per-tile prefix launch after the first two (the first pair -- trans and its prefix -- only when the forward did
//      not leave the transmittances in fwd.seg_T, as the autograd path asks it to): surfel_bwd_trans_kernel, surfel_bwd_prefix_T_kernel,
//      surfel_bwd_sums_kernel, surfel_bwd_prefix_sums_kernel, surfel_bwd_grad_kernel.  Per contributing (pixel, entry) pair
//          v_i         = gC.c_i + gN.n_i + gD d_i + gDist (m_i^2 W - 2 m_i M1 + M2)     (dist = sum_{j<i} w_i w_j (m_i - m_j)^2)
//          dL/dalpha_i = T_i v_i - (V - P_i) / (1 - alpha_i) - T_final (gC.bg - gA) / (1 - alpha_i),   P_i = sum_{j<=i} w_j v_j
//          dL/ddepth_i = w_i gD + 2 gDist w_i (m_i W - M1) dm/dd
//      and from them the gradients of opacity, colour, normal, centre and of Tu / Tv / Tw through
//      p = (px Tw - Tu) x (py Tw - Tv); the 18 gradient words of an entry are accumulated in LDS per segment and flushed
//      with one global atomic per word and (segment, entry);
//   4. surfel_preprocess_bwd_kernel  per Gaussian, summed over the views in registers: through M = Hm P N_pix (and the bounding-box
//      centre formula for the low-pass filter's centre), the view rotation of the normal and the normalised quaternion to means3D,
//      scales, rotations, opacities and colours.
#include <hip/hip_fp16.h>

#include "surfel_common.h"

namespace ga {

constexpr int kBRec = 24;    // backward record floats: Tu(3) Tv(3) Tw(3) xy(2) opa nv(3) rgb(3) | cull half-extents rx ry | pad
constexpr int kGRec = 18;    // gradient record floats: dTu(3) dTv(3) dTw(3) dxy(2) dopa dnv(3) drgb(3)
constexpr int kBwdChunk = 128;
#ifndef GA_BWD_WAVE_MAJOR_LANES
#define GA_BWD_WAVE_MAJOR_LANES 12   // a wave walks entry-major when its entries are evaluated by this many of its lanes on average
#endif

struct SplatFwd {   // the forward's per-splat quantities (surfel_preprocess.hip / oracle_preprocess, same formulas)
    float Tu[3], Tv[3], Tw[3], cx, cy, nv[3], mult;
    float tu[3], tv[3], nn[3], su, sv, qs, q[4];   // rotation columns, scaled axes, 1 / |q|
    float Dn;                                      // sum t Tw^2 (bounding-box denominator)
};

__device__ __forceinline__ void splat_forward(const float *__restrict__ means3D, const float *__restrict__ scales,
                                              const float *__restrict__ rotations, const float *__restrict__ vm,
                                              const float *__restrict__ pm, float scale_modifier, const Dims &dm, int i, SplatFwd &o)
{
    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    const float q0 = rotations[4 * i], q1 = rotations[4 * i + 1], q2 = rotations[4 * i + 2], q3 = rotations[4 * i + 3];
    o.q[0] = q0; o.q[1] = q1; o.q[2] = q2; o.q[3] = q3;
    o.qs = 1.0f / sqrtf(((q3 * q3 + q0 * q0) + q1 * q1) + q2 * q2);
    const float r = q0 * o.qs, x = q1 * o.qs, y = q2 * o.qs, z = q3 * o.qs;
    o.tu[0] = 1.f - 2.f * (y * y + z * z); o.tu[1] = 2.f * (x * y + r * z); o.tu[2] = 2.f * (x * z - r * y);
    o.tv[0] = 2.f * (x * y - r * z); o.tv[1] = 1.f - 2.f * (x * x + z * z); o.tv[2] = 2.f * (y * z + r * x);
    o.nn[0] = 2.f * (x * z + r * y); o.nn[1] = 2.f * (y * z - r * x); o.nn[2] = 1.f - 2.f * (x * x + y * y);
    o.su = scale_modifier * scales[2 * i]; o.sv = scale_modifier * scales[2 * i + 1];
    const float halfW = (float)dm.W / 2.0f, halfH = (float)dm.H / 2.0f;
    const float cW = (float)(dm.W - 1) / 2.0f, cH = (float)(dm.H - 1) / 2.0f;
    const float Hm[3][3] = {{o.tu[0] * o.su, o.tu[1] * o.su, o.tu[2] * o.su}, {o.tv[0] * o.sv, o.tv[1] * o.sv, o.tv[2] * o.sv}, {px, py, pz}};
    float M[3][3];
    for (int a = 0; a < 3; ++a) {
        float A[4];
        for (int j = 0; j < 4; ++j) {
            float s = Hm[a][0] * pm[0 + j] + Hm[a][1] * pm[4 + j] + Hm[a][2] * pm[8 + j];
            if (a == 2) s = s + pm[12 + j];
            A[j] = s;
        }
        M[a][0] = A[0] * halfW + A[3] * cW;
        M[a][1] = A[1] * halfH + A[3] * cH;
        M[a][2] = A[3];
    }
    for (int a = 0; a < 3; ++a) { o.Tu[a] = M[a][0]; o.Tv[a] = M[a][1]; o.Tw[a] = M[a][2]; }
    const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    float nvx = vm[0] * o.nn[0] + vm[4] * o.nn[1] + vm[8] * o.nn[2];
    float nvy = vm[1] * o.nn[0] + vm[5] * o.nn[1] + vm[9] * o.nn[2];
    float nvz = vm[2] * o.nn[0] + vm[6] * o.nn[1] + vm[10] * o.nn[2];
    const float cs = -((vx * nvx + vy * nvy) + vz * nvz);
    o.mult = cs > 0.0f ? 1.0f : -1.0f;
    o.nv[0] = o.mult * nvx; o.nv[1] = o.mult * nvy; o.nv[2] = o.mult * nvz;
    const float t[3] = {kCutoff * kCutoff, kCutoff * kCutoff, -1.0f};
    o.Dn = t[0] * o.Tw[0] * o.Tw[0] + t[1] * o.Tw[1] * o.Tw[1] + t[2] * o.Tw[2] * o.Tw[2];
    const float inv = 1.0f / o.Dn;
    o.cx = inv * (t[0] * o.Tu[0] * o.Tw[0] + t[1] * o.Tu[1] * o.Tw[1] + t[2] * o.Tu[2] * o.Tw[2]);
    o.cy = inv * (t[0] * o.Tv[0] * o.Tw[0] + t[1] * o.Tv[1] * o.Tw[1] + t[2] * o.Tv[2] * o.Tw[2]);
}

constexpr int kBwdSeg = 128;   // list entries per segment (= kBwdChunk, the LDS image of the segment kernels)

struct BwdPlan {   // the segment table and the per-(segment, pixel) exchange arrays, all inside `scratch`
    uint32_t *seg_base;     // [V tiles + 1] first segment of every tile's list; the last word is the number of segments
    uint32_t *seg_owner;    // [max_segs] the (view, tile) a segment belongs to
    float *Tseg;            // [max_segs][256]  trans: T_seg; after the first prefix launch: T_start
    float *Tend;            // [max_segs][256]
    float4 *part;           // [max_segs][256]  sums: W', M1', M2', A'; after the second prefix launch: their sums over the earlier segments
    float4 *total;          // [V tiles][256]   W, M1, M2, sum A' of the whole list
    float *Tfinal;          // [V tiles][256]
    uint32_t *big;          // [1 + 2 max_segs] number and list of the segments whose pairs exceed the gradient kernel's pair table,
                            // then one word per segment: 1 = it is on that list (both written by the sums launch)
    uint32_t max_segs;
};

// The segment table, in closed form: the segments of tile vt are numbered from floor(tile_start[vt] / 128) + vt -- increasing with vt,
// never overlapping (floor((a + n) / 128) - floor(a / 128) >= ceil(n / 128) - 1), at most one unused number between two tiles, all
// below capacity / 128 + tiles -- so every tile fills in its own rows and no scan over the tiles is needed (round 2: one
// workgroup, 27 us of serial scan).  One thread per tile.
__device__ __forceinline__ void bwd_segtable(const uint32_t *__restrict__ tile_start, int vtiles, const BwdPlan &pl,
                                             const int64_t *__restrict__ status, int vt)
{
    const bool overflow = status[GA_STATUS_OVERFLOW] != 0;
    if (vt == 0) {
        pl.big[0] = 0;
        pl.seg_base[vtiles] = overflow ? 0u : tile_start[vtiles] / kBwdSeg + (uint32_t)vtiles;   // (one past the last number in use)
    }
    if (overflow || vt >= vtiles) return;
    const uint32_t beg = tile_start[vt], end = tile_start[vt + 1];
    const uint32_t first = beg / kBwdSeg + (uint32_t)vt, n = (end - beg + kBwdSeg - 1) / kBwdSeg;
    const uint32_t next_first = end / kBwdSeg + (uint32_t)vt + 1u;
    pl.seg_base[vt] = first;
    for (uint32_t j = first; j < next_first; ++j)
        if (j < pl.max_segs) pl.seg_owner[j] = j < first + n ? (uint32_t)vt : 0xffffffffu;
}

// (grid row V builds the segment table, one thread per tile)
__global__ __launch_bounds__(1024) void surfel_bwd_record_kernel(const float *__restrict__ means3D, const float *__restrict__ opacities,
                                                                 const float *__restrict__ colors, const float *__restrict__ scales,
                                                                 const float *__restrict__ rotations, const float *__restrict__ viewmatrix,
                                                                 const float *__restrict__ projmatrix, float scale_modifier, Dims dm,
                                                                 const int32_t *__restrict__ radii, const float *__restrict__ fwd_record,
                                                                 float *__restrict__ brec, float *__restrict__ grec,
                                                                 const uint32_t *__restrict__ tile_start, BwdPlan pl,
                                                                 const int64_t *__restrict__ status)
{
    if ((int)blockIdx.y == dm.V) {
        for (int vt = (int)(blockIdx.x * 1024 + threadIdx.x); vt < max(dm.V * dm.tiles, 1); vt += (int)(gridDim.x * 1024))
            bwd_segtable(tile_start, dm.V * dm.tiles, pl, status, vt);
        return;
    }
    const int v = blockIdx.y, i = blockIdx.x * 1024 + threadIdx.x;
    if (i >= dm.N) return;
    const size_t idx = (size_t)v * dm.N + i;
    float *g = grec + idx * kGRec;
    for (int f = 0; f < kGRec; ++f) g[f] = 0.0f;
    if (radii[idx] <= 0) return;
    SplatFwd s;
    splat_forward(means3D, scales, rotations, viewmatrix + 16 * v, projmatrix + 16 * v, scale_modifier, dm, i, s);
    float *b = brec + idx * kBRec;
    for (int a = 0; a < 3; ++a) { b[a] = s.Tu[a]; b[3 + a] = s.Tv[a]; b[6 + a] = s.Tw[a]; b[12 + a] = s.nv[a]; b[15 + a] = colors[3 * i + a]; }
    b[9] = s.cx; b[10] = s.cy; b[11] = opacities[i];
    // the forward's conservative {alpha >= 1/255} box (two fp16 half-extents about the centre, surfel_common.h): pairs outside
    // it were not evaluated by the forward and contribute nothing here either
    const uint32_t cull = __float_as_uint(fwd_record[idx * kRec + 15]);
    b[18] = __half2float(__ushort_as_half((unsigned short)(cull & 0xffffu)));
    b[19] = __half2float(__ushort_as_half((unsigned short)(cull >> 16)));
}

// one (pixel, entry) evaluation: the forward's arithmetic (oracle_blend) with what the gradient needs kept.  The record is
// read from the workgroup's LDS image, one plane per field (b[f * kBwdChunk]), so that lanes on different entries hit
// different banks.
struct PairFwd {
    float k[3], l[3], p[3], Tw[3], rz, sx, sy, rho, G, raw, alpha, depth, dxc, dyc;
    bool use3d, ok;
};

#define GA_BF(f) b[(f) * kBwdChunk]

// (reciprocal and exponential as the forward kernel takes them: v_rcp_f32 / v_exp_f32, surfel_blend.hip)
__device__ __forceinline__ void pair_forward(const float *__restrict__ b, float pxf, float pyf, PairFwd &o)
{
    o.ok = false;
    for (int a = 0; a < 3; ++a) {
        o.Tw[a] = GA_BF(6 + a);
        o.k[a] = pxf * o.Tw[a] - GA_BF(a);
        o.l[a] = pyf * o.Tw[a] - GA_BF(3 + a);
    }
    o.p[0] = o.k[1] * o.l[2] - o.k[2] * o.l[1];
    o.p[1] = o.k[2] * o.l[0] - o.k[0] * o.l[2];
    o.p[2] = o.k[0] * o.l[1] - o.k[1] * o.l[0];
    if (o.p[2] == 0.0f) return;
    o.rz = __builtin_amdgcn_rcpf(o.p[2]);
    o.sx = o.p[0] * o.rz; o.sy = o.p[1] * o.rz;
    const float rho3d = o.sx * o.sx + o.sy * o.sy;
    o.dxc = GA_BF(9) - pxf; o.dyc = GA_BF(10) - pyf;
    const float rho2d = kFilterInvSquare * (o.dxc * o.dxc + o.dyc * o.dyc);
    o.use3d = rho3d <= rho2d;
    o.rho = fminf(rho3d, rho2d);
    o.depth = o.use3d ? (o.sx * o.Tw[0] + o.sy * o.Tw[1]) + o.Tw[2] : o.Tw[2];
    if (o.depth < kNear) return;
    if (o.rho < 0.0f) return;
    o.G = __builtin_amdgcn_exp2f(o.rho * -0.72134752044f);
    o.raw = GA_BF(11) * o.G;
    o.alpha = fminf(0.99f, o.raw);
    if (o.alpha < 1.0f / 255.0f) return;
    o.ok = true;
}

// ---- the blend backward, one workgroup per 128-entry SEGMENT of a tile's list -------------------------------------------
// A tile's list is cut into segments of kBwdSeg entries and every segment is a workgroup of its own in each of three
// launches, so that the longest list of a view (thousands of entries) is no longer one workgroup's serial walk:
//   trans  T_seg(pixel) = prod (1 - alpha) over the segment's contributing entries;
//   (a one-workgroup-per-tile launch after each of the first two turns T_seg into T_start and the sums into prefixes and totals)
//   sums   with T_start = prod of the earlier segments' T_seg (a pixel whose T_start is below 1e-4 terminated earlier:
//          the test that ends a walk fires at the entry that would take T below 1e-4, and T only falls): the forward walk
//          of the segment with its termination, leaving W' = sum w, M1' = sum w m, M2' = sum w m^2,
//          A' = sum w (gC.c + gN.n + gD d) and the transmittance the pixel leaves the segment with;
//   grad   totals and prefixes from the tile's segments (W, M1, M2, V = sum A' + 2 gDist (W M2 - M1^2), the final
//          transmittance, P_start = sum over earlier segments of A' + gDist (W M2' - 2 M1 M1' + M2 W')), then the
//          gradient walk of the segment.
// Shared by the three: the segment's records in LDS by field plane, and per 64-entry half the survivor masks -- bit e of
// col[h][c] says that column c of the tile lies inside entry e's cull box, row[h][r] likewise; a pixel's survivors are
// col & row, so a lane walks only the entries the forward evaluated for its pixel (4 % of all pairs at BASELINE configs[1]).
static_assert(kBwdSeg == kBwdChunk, "one LDS image per segment");
#ifndef GA_BWD_ABLATE
#define GA_BWD_ABLATE 0   // timing-only builds (wrong results): 1 no tail atomics, 2 no phase B, 4 everything takes the former walk, 8 no phase A
#endif
#ifndef GA_BWD_PAIR_CAP
#define GA_BWD_PAIR_CAP 2528
#endif
constexpr int kPairCap = GA_BWD_PAIR_CAP;   // slots of the gradient kernel's pair table; segments with more pairs (large splats) go entry-major

struct BwdShared {
    float rec[20][kBwdChunk];        // field planes 0 .. 19 of the 24-float records (the rest is padding)
    unsigned long long col[2][kTile], row[2][kTile];
    unsigned long long colq[2][2], rowq[2][2];    // union of the masks of the columns / rows 0..7 and 8..15 (a wave = an 8 x 8 quadrant)
    uint32_t id[kBwdChunk];
    uint32_t wave_area[2];   // survivor pairs of entries 0..63 / 64..127 (sum of the cull rectangles' areas)
};


// Sums of 18 words over the wavefront in 5 registers: halves of the wave exchanged between pairs of words
// (v_permlane32_swap: one add then carries two words, 32 lanes each), rows of 16 lanes between pairs of those
// (v_permlane16_swap: four words per register, one per row), then an inclusive scan along each row (DPP).  The totals end
// in the last lane of every row: of word 4 i + {0, 2, 1, 3}[row] in q[i] (i < 4), of word 16 (rows 0, 1) / 17 (rows 2, 3) in
// q[4].  All 64 lanes must be active.
// (inline assembly: the second result of __builtin_amdgcn_permlane{16,32}_swap comes back as a copy of the first with this
// compiler -- tools/swap_test.hip; the s_nop pairs cover the VALU-write -> swap-read and swap-write -> VALU-read
// wait states, which the hazard recogniser does not see through an asm statement)
__device__ __forceinline__ void swap_halves(float &a, float &b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap_rows(float &a, float &b)
{
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
template <int kCtrl>
__device__ __forceinline__ float dpp_row(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), kCtrl, 0xf, 0xf, true));
}
__device__ __forceinline__ void wave_totals18(const float (&w)[18], float (&q)[5])
{
    float r[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { float a = w[2 * j], b = w[2 * j + 1]; swap_halves(a, b); r[j] = a + b; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { float a = r[2 * i], b = r[2 * i + 1]; swap_rows(a, b); q[i] = a + b; }
    { float a = r[8], b = r[8]; asm("" : "+v"(b)); swap_rows(a, b); q[4] = a + b; }   // (two registers: the swap is in place)
#pragma unroll
    for (int i = 0; i < 5; ++i) q[i] += dpp_row<0x111>(q[i]);   // row_shr:1
#pragma unroll
    for (int i = 0; i < 5; ++i) q[i] += dpp_row<0x112>(q[i]);   // row_shr:2
#pragma unroll
    for (int i = 0; i < 5; ++i) q[i] += dpp_row<0x114>(q[i]);   // row_shr:4
#pragma unroll
    for (int i = 0; i < 5; ++i) q[i] += dpp_row<0x118>(q[i]);   // row_shr:8
}

struct SegCtx {
    int vt, v, tx, ty, k, nseg;      // (view, tile), this segment's index within the tile's list, the list's segments
    uint32_t s0, cbeg, cn;           // first segment of the tile; this segment's entries
    int lx, ly, pxi, pyi;            // my pixel inside the tile / in the image
    bool inside;
    float pxf, pyf;
    size_t vbase;
};

__device__ __forceinline__ bool seg_context(const uint32_t *__restrict__ tile_start, const BwdPlan &pl, const Dims &dm, SegCtx &c,
                                            uint32_t s)
{
    const int vtiles = dm.V * dm.tiles;
    if (s >= pl.seg_base[vtiles] || s >= pl.max_segs) return false;
    if (pl.seg_owner[s] == 0xffffffffu) return false;   // (a number between two tiles that no segment has)
    c.vt = (int)pl.seg_owner[s];
    c.s0 = pl.seg_base[c.vt];
    c.k = (int)(s - c.s0);
    c.v = c.vt / dm.tiles;
    const int tile = c.vt - c.v * dm.tiles;
    c.tx = tile % dm.gx; c.ty = tile / dm.gx;
    const uint32_t beg = tile_start[c.vt], end = tile_start[c.vt + 1];
    c.nseg = (int)((end - beg + kBwdSeg - 1) / kBwdSeg);
    c.cbeg = beg + (uint32_t)c.k * kBwdSeg;
    c.cn = min((uint32_t)kBwdSeg, end - c.cbeg);
    // a wave is an 8 x 8 quadrant of the tile (fewer entries per wave than a 16 x 4 strip meets)
    c.lx = (int)(((threadIdx.x >> 6) & 1) * 8 + (threadIdx.x & 7));
    c.ly = (int)((threadIdx.x >> 7) * 8 + ((threadIdx.x >> 3) & 7));
    c.pxi = c.tx * kTile + c.lx;
    c.pyi = c.ty * kTile + c.ly;
    c.inside = c.pxi < dm.W && c.pyi < dm.H;
    c.pxf = (float)c.pxi; c.pyf = (float)c.pyi;
    c.vbase = (size_t)c.v * dm.N;
    return true;
}

// stage the segment: records to the field planes (kFields of them), survivor masks of my pixel
// (xbits / ybits, threads 0..127 only: the tile columns / rows inside the cull box of entry threadIdx.x -- an interval each)
template <int kFields>
__device__ __forceinline__ void stage_segment(BwdShared &sh, const SegCtx &c, const uint32_t *__restrict__ point_list,
                                              const float *__restrict__ brec, unsigned long long &m0, unsigned long long &m1,
                                              uint32_t *xbits = nullptr, uint32_t *ybits = nullptr)
{
    const int se = threadIdx.x & (kBwdChunk - 1), spart = threadIdx.x / kBwdChunk;   // entry, half of its record
    float bx = 0.f, by = 0.f, rx = -1.f, ry = -1.f;   // (an absent entry's box is empty)
    if ((uint32_t)se < c.cn) {
        const uint32_t id = point_list[c.cbeg + se];
        if (spart == 0) sh.id[se] = id;
        const float4 *src = reinterpret_cast<const float4 *>(brec + (c.vbase + id) * kBRec) + 3 * spart;
        if (spart == 0) {
            const float4 q0 = src[0], q1 = src[1], q2 = src[2];
            sh.rec[0][se] = q0.x; sh.rec[1][se] = q0.y; sh.rec[2][se] = q0.z; sh.rec[3][se] = q0.w;
            sh.rec[4][se] = q1.x; sh.rec[5][se] = q1.y; sh.rec[6][se] = q1.z; sh.rec[7][se] = q1.w;
            sh.rec[8][se] = q2.x; sh.rec[9][se] = q2.y; sh.rec[10][se] = q2.z; sh.rec[11][se] = q2.w;
            bx = q2.y; by = q2.z;
        } else {
            const float4 q1 = src[1];
            sh.rec[18][se] = q1.z; sh.rec[19][se] = q1.w;
            if (kFields > 12) {
                const float4 q0 = src[0];
                sh.rec[12][se] = q0.x; sh.rec[13][se] = q0.y; sh.rec[14][se] = q0.z; sh.rec[15][se] = q0.w;
                sh.rec[16][se] = q1.x; sh.rec[17][se] = q1.y;
            }
        }
    }
    // waves 0,1 hold the centres of entries 0..63 / 64..127, waves 2,3 their half-extents: hand the half-extents over
    __syncthreads();
    if (spart == 0) {
        if ((uint32_t)se < c.cn) { rx = sh.rec[18][se]; ry = sh.rec[19][se]; }
        const int h = se >> 6;
        const float ox = bx - (float)(c.tx * kTile), oy = by - (float)(c.ty * kTile);
        unsigned long long cs = 0ull, rs = 0ull;
        uint32_t xb = 0, yb = 0;
        for (int q = 0; q < kTile; ++q) {
            const bool inx = fabsf((float)q - ox) <= rx, iny = fabsf((float)q - oy) <= ry;
            const unsigned long long mc = __ballot(inx);
            const unsigned long long mr = __ballot(iny);
            xb |= (uint32_t)inx << q; yb |= (uint32_t)iny << q;
            cs |= mc; rs |= mr;
            if ((se & 63) == 0) { sh.col[h][q] = mc; sh.row[h][q] = mr; }
            if ((q & 7) == 7) { if ((se & 63) == 0) { sh.colq[h][q >> 3] = cs; sh.rowq[h][q >> 3] = rs; } cs = 0ull; rs = 0ull; }
        }
        if (xbits) { *xbits = xb; *ybits = yb; }
    }
    __syncthreads();
    m0 = sh.col[0][c.lx] & sh.row[0][c.ly];
    m1 = sh.col[1][c.lx] & sh.row[1][c.ly];
}

__global__ __launch_bounds__(256) void surfel_bwd_trans_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                                                               const float *__restrict__ brec, Dims dm, BwdPlan pl)
{
    __shared__ BwdShared sh;
    SegCtx c;
    if (!seg_context(tile_start, pl, dm, c, blockIdx.x)) return;
    if (c.k == c.nseg - 1) return;   // nobody enters a later segment through the last one (a quarter of the segments at BASELINE configs[1])
    unsigned long long m[2];
    stage_segment<12>(sh, c, point_list, brec, m[0], m[1]);
    float T = 1.0f;
    for (int h = 0; h < 2; ++h) {
        unsigned long long mm = c.inside ? m[h] : 0ull;
        while (mm) {
            const int e = __builtin_ctzll(mm) + 64 * h;
            mm &= mm - 1;
            PairFwd f;
            pair_forward(&sh.rec[0][e], c.pxf, c.pyf, f);
            if (f.ok) T *= 1.0f - f.alpha;
        }
    }
    pl.Tseg[(size_t)blockIdx.x * 256 + threadIdx.x] = T;
}

// per tile (one workgroup, one thread per pixel), linear in the list's segments: T_seg -> T_start in place
__global__ __launch_bounds__(256) void surfel_bwd_prefix_T_kernel(const uint32_t *__restrict__ tile_start, BwdPlan pl, int vtiles)
{
    const int vt = blockIdx.x;
    if (pl.seg_base[vtiles] == 0) return;   // (no segments: nothing rendered, or the forward overflowed its lists)
    const uint32_t s0 = pl.seg_base[vt];
    const int nseg = (int)((tile_start[vt + 1] - tile_start[vt] + kBwdSeg - 1) / kBwdSeg);
    // (loads of four segments in flight at a time: the loop is a chain of dependent multiplies, not of dependent loads)
    float *__restrict__ base = pl.Tseg + (size_t)s0 * 256 + threadIdx.x;
    float T = 1.0f;
    int j = 0;
    for (; j + 4 < nseg; j += 4) {   // (the last segment's product is never formed: surfel_bwd_trans_kernel skips it)
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = base[(size_t)(j + u) * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) { base[(size_t)(j + u) * 256] = T; T *= t[u]; }
    }
    for (; j < nseg; ++j) {
        const float t = j + 1 < nseg ? base[(size_t)j * 256] : 1.0f;
        base[(size_t)j * 256] = T;
        T *= t;
    }
}

// ... and the sums of the earlier segments in place of every segment's own, the totals and the final transmittance (that of
// the last segment the pixel entered alive)
__global__ __launch_bounds__(256) void surfel_bwd_prefix_sums_kernel(const uint32_t *__restrict__ tile_start, BwdPlan pl, int vtiles)
{
    const int vt = blockIdx.x;
    if (pl.seg_base[vtiles] == 0) return;
    const uint32_t s0 = pl.seg_base[vt];
    const int nseg = (int)((tile_start[vt + 1] - tile_start[vt] + kBwdSeg - 1) / kBwdSeg);
    if (nseg == 0) return;
    float4 run = make_float4(0.f, 0.f, 0.f, 0.f);
    float T_final = 1.0f;
    const float *__restrict__ ts = pl.Tseg + (size_t)s0 * 256 + threadIdx.x;
    const float *__restrict__ te = pl.Tend + (size_t)s0 * 256 + threadIdx.x;
    float4 *__restrict__ pp = pl.part + (size_t)s0 * 256 + threadIdx.x;
    int j = 0;
    for (; j + 4 <= nseg; j += 4) {   // (four segments' loads in flight at a time)
        float a[4], b[4];
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = ts[(size_t)(j + u) * 256]; b[u] = te[(size_t)(j + u) * 256]; q[u] = pp[(size_t)(j + u) * 256]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (a[u] >= 0.0001f) T_final = b[u];
            pp[(size_t)(j + u) * 256] = run;
            run.x += q[u].x; run.y += q[u].y; run.z += q[u].z; run.w += q[u].w;
        }
    }
    for (; j < nseg; ++j) {
        if (ts[(size_t)j * 256] >= 0.0001f) T_final = te[(size_t)j * 256];
        const float4 q = pp[(size_t)j * 256];
        pp[(size_t)j * 256] = run;
        run.x += q.x; run.y += q.y; run.z += q.z; run.w += q.w;
    }
    pl.total[(size_t)vt * 256 + threadIdx.x] = run;
    pl.Tfinal[(size_t)vt * 256 + threadIdx.x] = T_final;
}

struct PixelGrads { float gC[3], gN[3], gD, gA, gDist, gMed; };

__device__ __forceinline__ void load_pixel_grads(const float *__restrict__ g_color, const float *__restrict__ g_others, const Dims &dm,
                                                 const SegCtx &c, PixelGrads &o)
{
    for (int q = 0; q < 3; ++q) { o.gC[q] = 0.f; o.gN[q] = 0.f; }
    o.gD = 0.f; o.gA = 0.f; o.gDist = 0.f; o.gMed = 0.f;
    if (!c.inside) return;
    const size_t HW = (size_t)dm.H * dm.W, pid = (size_t)c.pyi * dm.W + c.pxi;
    for (int q = 0; q < 3; ++q) o.gC[q] = g_color[((size_t)c.v * 3 + q) * HW + pid];
    const float *go = g_others + (size_t)c.v * 7 * HW + pid;
    o.gD = go[0]; o.gA = go[HW]; o.gN[0] = go[2 * HW]; o.gN[1] = go[3 * HW]; o.gN[2] = go[4 * HW]; o.gMed = go[5 * HW]; o.gDist = go[6 * HW];
}

__global__ __launch_bounds__(256) void surfel_bwd_sums_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                                                              const float *__restrict__ brec, Dims dm, BwdPlan pl,
                                                              const float *__restrict__ g_color, const float *__restrict__ g_others)
{
    __shared__ BwdShared sh;
    SegCtx c;
    if (!seg_context(tile_start, pl, dm, c, blockIdx.x)) return;
    const float kM = kFar / (kFar - kNear);
    float T = pl.Tseg[(size_t)blockIdx.x * 256 + threadIdx.x];   // (T_start since the prefix launch)
    bool done = !c.inside || T < 0.0001f;
    float W = 0.f, M1 = 0.f, M2 = 0.f, A = 0.f;
    const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (!__syncthreads_and(done)) {
        PixelGrads pg;
        load_pixel_grads(g_color, g_others, dm, c, pg);
        unsigned long long m[2];
        uint32_t xb = 0, yb = 0;
        stage_segment<kBRec>(sh, c, point_list, brec, m[0], m[1], &xb, &yb);
        // which gradient kernel takes this segment: the pairs the forward evaluated (entry by entry a rectangle of the tile)
        // against the pair table of surfel_bwd_grad_kernel
        if (threadIdx.x < kBwdChunk) {
            uint32_t area = (uint32_t)(__popc(xb) * __popc(yb));
            for (int o = 32; o > 0; o >>= 1) area += __shfl_xor(area, o, 64);
            if ((threadIdx.x & 63) == 0) sh.wave_area[threadIdx.x >> 6] = area;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool big = sh.wave_area[0] + sh.wave_area[1] > (uint32_t)kPairCap || (GA_BWD_ABLATE & 4);
            pl.big[1 + pl.max_segs + blockIdx.x] = big ? 1u : 0u;
            if (big) pl.big[1 + atomicAdd(pl.big, 1u)] = blockIdx.x;
        }
        for (int h = 0; h < 2; ++h) {
            unsigned long long mm = done ? 0ull : m[h];
            while (mm) {
                const int e = __builtin_ctzll(mm) + 64 * h;
                mm &= mm - 1;
                const float *b = &sh.rec[0][e];
                PairFwd f;
                pair_forward(b, c.pxf, c.pyf, f);
                if (!f.ok) continue;
                const float test_T = T * (1.0f - f.alpha);
                if (test_T < 0.0001f) { done = true; break; }
                const float w = f.alpha * T, mz = kM * (1.0f - kNear * __builtin_amdgcn_rcpf(f.depth));
                W += w; M1 += mz * w; M2 += mz * mz * w;
                A += w * ((pg.gC[0] * GA_BF(15) + pg.gC[1] * GA_BF(16) + pg.gC[2] * GA_BF(17)) +
                          (pg.gN[0] * GA_BF(12) + pg.gN[1] * GA_BF(13) + pg.gN[2] * GA_BF(14)) + pg.gD * f.depth);
                T = test_T;
            }
        }
    }
    pl.part[slot] = make_float4(W, M1, M2, A);
    pl.Tend[slot] = T;
}

// ---- the gradient walk ---------------------------------------------------------------------------------------------------
// Every contributing (pixel, entry) pair has 18 gradient words that must be summed per entry.  LDS float atomics cannot carry
// that: ds_add_f32 is executed one lane at a time on gfx950 (tools/lds_atomic_rate.hip: 52 CU cycles for a 64-lane
// instruction on 64 different addresses, 0.8 per active lane, against 1.4 cycles for ds_add_u32), and with one lane-major walk
// adding its 18 words per pair they were 510 of the kernel's 810 us at BASELINE configs[1] (in-situ ablation).  So the sums
// are formed in registers and the kernel has two phases (PAIR-MAJOR path, whenever the segment's pairs fit its LDS table):
//   A  lanes = pixels, each walks its own survivors in list order -- the part that is sequential per pixel -- and leaves per
//      pair only T_i (transmittance on entering the pair) and P_i (the running sum of w v up to and including it) in a pair
//      table.  A pixel's survivors of entry e are exactly the pixels of e's cull box (an axis-aligned rectangle of the tile:
//      the masks are products of a column and a row interval), so the table is laid out entry by entry, rectangle row-major:
//      slot = base[e] + (y - y0) w + (x - x0), base = exclusive prefix of the rectangle areas.
//   B  lanes = table slots (256 per round: no idle lanes whatever the spread of the list lengths).  A lane re-evaluates its pair
//      from the record and its pixel's constants (LDS), takes T_i and P_i from the table and forms the 18 words; the slots of
//      an entry are consecutive lanes, so a segmented sum along each row of 16 lanes (4 DPP steps per word) leaves every
//      entry's total in the last lane of its run, which adds it to the global gradient record (~1.6 runs per entry).
// Segments with more than kPairCap pairs (large splats; 19.7 pairs per entry on average) keep the former walk with its LDS
// gradient image, in a kernel of their own (walking such a segment pair-major in several fills of the table was built and
// measured: slower on the stress scene -- the table has a slot for every survivor, also those of pixels that are finished,
// and with long lists behind saturated pixels that is most of them):
//   lane-major   every lane walks its own survivors and adds its 18 words to the segment's gradient image with LDS atomics;
//   entry-major  the wave walks the union of its lanes' survivors, all lanes evaluate the same entry (one broadcast record
//                read), the words are summed across the wave (wave_totals18) and four lanes add the totals: cheaper as soon
//                as an entry is evaluated by GA_BWD_WAVE_MAJOR_LANES lanes of the wave on average.
constexpr int kTailBatch = 8;
static_assert(kGRec == 18 && kTailBatch * kGRec <= 144, "(i * 3641) >> 16 == i / 18 was checked for i < 144");
constexpr int kPixFields = 15;   // gC(3) gN(3) gD gDist | W M1 M2 Vtot | T_final (gC.bg - gA) | gMedian T_final

struct PairShared {
    float pix[kPixFields][256];       // per-pixel constants, pixel = ly * 16 + lx
    float pT[kPairCap], pP[kPairCap];
    uint32_t desc[kBwdChunk];         // per entry: base | x0 << 16 | y0 << 20 | w << 24 (w = 0: empty box)
    unsigned char pE[kPairCap];       // entry of a slot
    uint32_t wave_area[2];
    float4 tails[4][kTailBatch][5];      // per wave: run totals (18 words in 20) on their way to the global record
    uint32_t tail_id[4][kTailBatch];
};

struct PixC { float gC[3], gN[3], gD, gDist, W, M1, M2, Vtot, Tfb, gMed, Tfin; };
struct PairV { float w, mz, rd, vi; };

__device__ __forceinline__ PairV pair_value(const float *__restrict__ b, const PairFwd &f, float T, const PixC &pc)
{
    const float kM = kFar / (kFar - kNear);
    PairV o;
    o.rd = __builtin_amdgcn_rcpf(f.depth);
    o.w = f.alpha * T;
    o.mz = kM * (1.0f - kNear * o.rd);
    const float Dq = o.mz * o.mz * pc.W - 2.0f * o.mz * pc.M1 + pc.M2;
    o.vi = (pc.gC[0] * GA_BF(15) + pc.gC[1] * GA_BF(16) + pc.gC[2] * GA_BF(17)) +
           (pc.gN[0] * GA_BF(12) + pc.gN[1] * GA_BF(13) + pc.gN[2] * GA_BF(14)) + pc.gD * f.depth + pc.gDist * Dq;
    return o;
}

// the 18 gradient words of one contributing pair; P = sum of w v up to and including this pair
__device__ __forceinline__ void pair_words(const PairFwd &f, const PairV &v, float T, float P, const PixC &pc, float pxf, float pyf,
                                           float *cw)
{
    const float kM = kFar / (kFar - kNear);
    const float inv1ma = __builtin_amdgcn_rcpf(1.0f - f.alpha);
    const float dL_dalpha = T * v.vi - (pc.Vtot - P) * inv1ma - pc.Tfb * inv1ma;
    // The median depth (allmap channel 5) is the depth of the LAST contributing pair entered with T > 0.5: the next one is
    // entered with T (1 - alpha) <= 0.5, or there is none -- then the pixel's final transmittance is this product (up to the
    // rounding of the per-segment products; any later contributing pair would have taken at least 1/255 off it).  Which
    // pair that is counts as a constant of the gradient.
    const float next_T = T * (1.0f - f.alpha);
    const bool is_median = T > 0.5f && (!(next_T > 0.5f) || pc.Tfin > 0.998f * next_T);
    const float dL_ddepth = v.w * pc.gD + 2.0f * pc.gDist * v.w * (v.mz * pc.W - pc.M1) * (kM * kNear * v.rd * v.rd) +
                            (is_median ? pc.gMed : 0.0f);
    for (int q = 0; q < 3; ++q) { cw[15 + q] = v.w * pc.gC[q]; cw[12 + q] = v.w * pc.gN[q]; }
    const float dL_draw = f.raw > 0.99f ? 0.0f : dL_dalpha;       // the clamp carries no gradient
    cw[11] = f.G * dL_draw;
    const float dL_drho = -0.5f * f.raw * dL_draw;
    float dsx = 0.f, dsy = 0.f, dTw[3] = {0.f, 0.f, dL_ddepth};
    cw[9] = 0.f; cw[10] = 0.f;
    if (f.use3d) {
        dsx = 2.0f * f.sx * dL_drho + dL_ddepth * f.Tw[0];
        dsy = 2.0f * f.sy * dL_drho + dL_ddepth * f.Tw[1];
        dTw[0] = dL_ddepth * f.sx; dTw[1] = dL_ddepth * f.sy;
    } else {
        cw[9] = 2.0f * kFilterInvSquare * f.dxc * dL_drho;
        cw[10] = 2.0f * kFilterInvSquare * f.dyc * dL_drho;
    }
    // s = p.xy / p.z ; p = k x l ; k = px Tw - Tu ; l = py Tw - Tv
    const float ipz = f.rz;
    const float gp[3] = {dsx * ipz, dsy * ipz, -(dsx * f.sx + dsy * f.sy) * ipz};
    const float dk[3] = {f.l[1] * gp[2] - f.l[2] * gp[1], f.l[2] * gp[0] - f.l[0] * gp[2], f.l[0] * gp[1] - f.l[1] * gp[0]};
    const float dl[3] = {gp[1] * f.k[2] - gp[2] * f.k[1], gp[2] * f.k[0] - gp[0] * f.k[2], gp[0] * f.k[1] - gp[1] * f.k[0]};
    for (int q = 0; q < 3; ++q) {
        cw[q] = -dk[q];
        cw[3 + q] = -dl[q];
        cw[6 + q] = dTw[q] + pxf * dk[q] + pyf * dl[q];
    }
}

template <int kCtrl>
__device__ __forceinline__ int dpp_row_i(int v)
{
    return __builtin_amdgcn_mov_dpp(v, kCtrl, 0xf, 0xf, true);
}
// One step of the segmented inclusive sum along a row of 16 lanes: lanes whose neighbour kShr to the left holds the same key
// add its partial sums (keys are runs of equal values; a lane shifted in from outside the row reads key 0 and value 0).
// One v_fmac_f32 with a DPP source per word: acc += neighbour(acc) * same, same = 1.0 / 0.0 (the values are finite).
// (inline assembly, one statement per step: hipcc does not fold the DPP move into the multiply-add.  Inside the statement
// every word is read 19 instructions after the previous step wrote it; the s_nop in front covers the two wait states a DPP
// read needs after a VALU write for whatever the compiler placed before the statement -- the hazard recogniser does not look
// inside it)
#define GA_SEG_STEP(SHR)                                                                                     \
    do {                                                                                                    \
        const float same_ = dpp_row_i<0x110 + SHR>(key) == key ? 1.0f : 0.0f;                               \
        asm("s_nop 1\n\t"                                                                                   \
            "v_fmac_f32_dpp %0, %0, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %1, %1, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %2, %2, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %3, %3, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %4, %4, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %5, %5, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %6, %6, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %7, %7, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %8, %8, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %9, %9, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"        \
            "v_fmac_f32_dpp %10, %10, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            "v_fmac_f32_dpp %11, %11, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            "v_fmac_f32_dpp %12, %12, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            "v_fmac_f32_dpp %13, %13, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            "v_fmac_f32_dpp %14, %14, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            "v_fmac_f32_dpp %15, %15, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            "v_fmac_f32_dpp %16, %16, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            "v_fmac_f32_dpp %17, %17, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            "v_fmac_f32_dpp %18, %18, %19 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
            : "+v"(cw[0]), "+v"(cw[1]), "+v"(cw[2]), "+v"(cw[3]), "+v"(cw[4]), "+v"(cw[5]), "+v"(cw[6]), "+v"(cw[7]), "+v"(cw[8]), "+v"(cw[9]), "+v"(cw[10]), "+v"(cw[11]), "+v"(cw[12]), "+v"(cw[13]), "+v"(cw[14]), "+v"(cw[15]), "+v"(cw[16]), "+v"(cw[17]), "+v"(cnt) \
            : "v"(same_));                                                                                  \
    } while (0)
static_assert(kGRec == 18, "GA_SEG_STEP names the 18 words and the count");
#define GA_LDS_ORDER() asm volatile("" ::: "memory")   // DS operations of one wave reach the LDS in program order

__global__ __launch_bounds__(256) void surfel_bwd_grad_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                                                              const float *__restrict__ brec, const float *__restrict__ bg, Dims dm, BwdPlan pl,
                                                              const float *__restrict__ g_color, const float *__restrict__ g_others,
                                                              float *__restrict__ grec)
{
    __shared__ BwdShared sh;
    __shared__ PairShared pr;
    SegCtx c;
    if (!seg_context(tile_start, pl, dm, c, blockIdx.x)) return;
    // (written by the sums launch for every segment it walked; a segment it skipped -- every pixel finished -- is skipped here as well)
    if (pl.big[1 + pl.max_segs + blockIdx.x] != 0u) return;
    // my pixel's state on entering the segment, and the totals of its whole walk
    const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x;
    float T = pl.Tseg[slot];
    bool done = !c.inside || T < 0.0001f;
    if (__syncthreads_and(done)) return;
    const float4 pre = pl.part[slot], tot = pl.total[(size_t)c.vt * 256 + threadIdx.x];
    const float T_final = pl.Tfinal[(size_t)c.vt * 256 + threadIdx.x];
    PixC pc;
    {
        PixelGrads pg;
        load_pixel_grads(g_color, g_others, dm, c, pg);
        for (int q = 0; q < 3; ++q) { pc.gC[q] = pg.gC[q]; pc.gN[q] = pg.gN[q]; }
        pc.gD = pg.gD; pc.gDist = pg.gDist;
        pc.W = tot.x; pc.M1 = tot.y; pc.M2 = tot.z;
        pc.Vtot = tot.w + 2.0f * pg.gDist * (tot.x * tot.z - tot.y * tot.y);
        pc.Tfb = T_final * ((pg.gC[0] * bg[0] + pg.gC[1] * bg[1] + pg.gC[2] * bg[2]) - pg.gA);
        pc.gMed = pg.gMed; pc.Tfin = T_final;
    }
    float P = pre.w + pc.gDist * (pc.W * pre.z - 2.0f * pc.M1 * pre.y + pc.M2 * pre.x);
    unsigned long long m[2];
    uint32_t xb = 0, yb = 0;
    stage_segment<kBRec>(sh, c, point_list, brec, m[0], m[1], &xb, &yb);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float pxf = c.pxf, pyf = c.pyf;

    // the entries' rectangles and their places in the pair table (threads 0..127 = entries)
    uint32_t area = 0, incl = 0;
    if (threadIdx.x < kBwdChunk) {
        area = (uint32_t)(__popc(xb) * __popc(yb));
        incl = area;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) pr.wave_area[wv] = incl;
    }
    __syncthreads();
    const uint32_t total = pr.wave_area[0] + pr.wave_area[1];
    if (total > (uint32_t)kPairCap) return;   // (cannot happen: those segments are on the other kernel's list)
    if (threadIdx.x < kBwdChunk) {
        const uint32_t base = incl - area + (wv == 1 ? pr.wave_area[0] : 0u);
        const uint32_t w = (uint32_t)__popc(xb);
        pr.desc[threadIdx.x] = base | (area ? ((uint32_t)__builtin_ctz(xb) << 16) | ((uint32_t)__builtin_ctz(yb) << 20) | (w << 24) : 0u);
        for (uint32_t q = 0; q < area; ++q) { pr.pE[base + q] = (unsigned char)threadIdx.x; pr.pT[base + q] = 0.0f; }
    }
    {
        const int pix = c.ly * kTile + c.lx;
        const float vals[kPixFields] = {pc.gC[0], pc.gC[1], pc.gC[2], pc.gN[0], pc.gN[1], pc.gN[2], pc.gD, pc.gDist,
                                        pc.W, pc.M1, pc.M2, pc.Vtot, pc.Tfb, pc.gMed, pc.Tfin};
#pragma unroll
        for (int q = 0; q < kPixFields; ++q) pr.pix[q][pix] = vals[q];
    }
    __syncthreads();
    const float tx0 = (float)(c.tx * kTile), ty0 = (float)(c.ty * kTile);
    {
        // A: my pixel's walk
        for (int h = 0; h < 2; ++h) {
            unsigned long long mm = (done || (GA_BWD_ABLATE & 8)) ? 0ull : m[h];
            while (mm) {
                const int e = __builtin_ctzll(mm) + 64 * h;
                mm &= mm - 1;
                const float *b = &sh.rec[0][e];
                PairFwd f;
                pair_forward(b, pxf, pyf, f);
                if (!f.ok) continue;
                const float test_T = T * (1.0f - f.alpha);
                if (test_T < 0.0001f) { done = true; break; }
                const PairV v = pair_value(b, f, T, pc);
                P += v.w * v.vi;
                const uint32_t d = pr.desc[e];
                const uint32_t s2 = (d & 0xffffu) + ((uint32_t)c.ly - ((d >> 20) & 15u)) * (d >> 24) + ((uint32_t)c.lx - ((d >> 16) & 15u));
                pr.pT[s2] = T;
                pr.pP[s2] = P;
                T = test_T;
            }
        }
        __syncthreads();
        if (GA_BWD_ABLATE & 2) return;
        // B: one lane per slot
        for (uint32_t p0 = 0; p0 < total; p0 += 256) {
            const uint32_t p = p0 + threadIdx.x;
            const bool valid = p < total;
            const int e = valid ? (int)pr.pE[p] : 0;
            const float Ti = valid ? pr.pT[p] : 0.0f;
            const bool on = Ti > 0.0f;     // (a contributing pair was entered with T >= 1e-4)
            float cw[kGRec];
#pragma unroll
            for (int q = 0; q < kGRec; ++q) cw[q] = 0.0f;
            if (__ballot(on) == 0ull) continue;     // (runs of entries behind the surface: nothing in these 64 slots)
            if (on) {
                const uint32_t d = pr.desc[e];
                const uint32_t r = p - (d & 0xffffu), w = d >> 24;
                // r / w for r < 256, w <= 16: (r + 0.5) / w is at least 1/32 away from an integer, v_rcp_f32 is good to 1 ulp
                const uint32_t yy = (uint32_t)(((float)r + 0.5f) * __builtin_amdgcn_rcpf((float)w));
                const uint32_t lx = ((d >> 16) & 15u) + (r - yy * w), ly = ((d >> 20) & 15u) + yy;
                const int pix = (int)(ly * kTile + lx);
                PixC q;
                q.gC[0] = pr.pix[0][pix]; q.gC[1] = pr.pix[1][pix]; q.gC[2] = pr.pix[2][pix];
                q.gN[0] = pr.pix[3][pix]; q.gN[1] = pr.pix[4][pix]; q.gN[2] = pr.pix[5][pix];
                q.gD = pr.pix[6][pix]; q.gDist = pr.pix[7][pix]; q.W = pr.pix[8][pix]; q.M1 = pr.pix[9][pix];
                q.M2 = pr.pix[10][pix]; q.Vtot = pr.pix[11][pix]; q.Tfb = pr.pix[12][pix];
                q.gMed = pr.pix[13][pix]; q.Tfin = pr.pix[14][pix];
                const float qx = tx0 + (float)lx, qy = ty0 + (float)ly;
                const float *b = &sh.rec[0][e];
                PairFwd f;
                pair_forward(b, qx, qy, f);    // (the same arithmetic on the same operands as in A: f.ok holds again)
                const PairV v = pair_value(b, f, Ti, q);
                pair_words(f, v, Ti, pr.pP[p], q, qx, qy, cw);
            }
            const int key = valid ? e + 1 : 0;
            float cnt = on ? 1.0f : 0.0f;       // (a 19th word: does the run hold a contributing pair at all)
            GA_SEG_STEP(1);
            GA_SEG_STEP(2);
            GA_SEG_STEP(4);
            GA_SEG_STEP(8);
            // the last lane of a run holds its totals.  One lane adding its 18 words to the global record is 18 separate
            // requests to the L2's atomic units (measured: 585 us per launch); so the totals go through a small per-wave LDS
            // buffer, kTailBatch runs at a time, and leave as one atomic per lane on consecutive words -- the coalesced form
            // the former per-segment flush has.
            const bool tail = valid && cnt > 0.0f && dpp_row_i<0x101>(key) != key;   // row_shl:1 (0 past the end of the row)
            const unsigned long long tm = __ballot(tail);
            const int nt = __popcll(tm);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0u));
            for (int b0 = 0; b0 < nt && !(GA_BWD_ABLATE & 1); b0 += kTailBatch) {
                if (tail && rank >= b0 && rank < b0 + kTailBatch) {
                    float4 *t4 = pr.tails[wv][rank - b0];
                    t4[0] = make_float4(cw[0], cw[1], cw[2], cw[3]); t4[1] = make_float4(cw[4], cw[5], cw[6], cw[7]);
                    t4[2] = make_float4(cw[8], cw[9], cw[10], cw[11]); t4[3] = make_float4(cw[12], cw[13], cw[14], cw[15]);
                    t4[4] = make_float4(cw[16], cw[17], 0.0f, 0.0f);
                    pr.tail_id[wv][rank - b0] = sh.id[e];
                }
                GA_LDS_ORDER();
                const int nw = min(kTailBatch, nt - b0) * kGRec;
                for (int i = lane; i < nw; i += 64) {
                    const int r = (i * 3641) >> 16, q = i - r * kGRec;      // i / 18 for i < 18 kTailBatch
                    const float val = reinterpret_cast<const float *>(pr.tails[wv][r])[q];
                    if (val != 0.0f) atomicAdd(grec + (c.vbase + pr.tail_id[wv][r]) * kGRec + q, val);
                }
                GA_LDS_ORDER();
            }
            if ((GA_BWD_ABLATE & 1) && tail) { float z = 0.f; for (int q = 0; q < kGRec; ++q) z += cw[q]; if (z == 123.4f) grec[0] = z; }
        }
    }
}

// The segments the pair table cannot hold (listed in pl.big by the kernel above): the walk with an LDS gradient image of the
// segment.  (A kernel of its own: its 24 KB of LDS allow four workgroups per CU, the pair table's 52 KB three.)
__global__ __launch_bounds__(256) void surfel_bwd_grad_big_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                                                                  const float *__restrict__ brec, const float *__restrict__ bg, Dims dm, BwdPlan pl,
                                                                  const float *__restrict__ g_color, const float *__restrict__ g_others,
                                                                  float *__restrict__ grec)
{
    __shared__ BwdShared sh;
    __shared__ float sgrad[kBwdChunk][kGRec + 1];   // (+1: the 18 words of neighbouring entries start in different banks)
    if (blockIdx.x >= pl.big[0]) return;
    const uint32_t seg = pl.big[1 + blockIdx.x];
    SegCtx c;
    if (!seg_context(tile_start, pl, dm, c, seg)) return;
    // my pixel's state on entering the segment, and the totals of its whole walk
    const size_t slot = (size_t)seg * 256 + threadIdx.x;
    float T = pl.Tseg[slot];
    bool done = !c.inside || T < 0.0001f;
    if (__syncthreads_and(done)) return;
    const float4 pre = pl.part[slot], tot = pl.total[(size_t)c.vt * 256 + threadIdx.x];
    const float T_final = pl.Tfinal[(size_t)c.vt * 256 + threadIdx.x];
    PixC pc;
    {
        PixelGrads pg;
        load_pixel_grads(g_color, g_others, dm, c, pg);
        for (int q = 0; q < 3; ++q) { pc.gC[q] = pg.gC[q]; pc.gN[q] = pg.gN[q]; }
        pc.gD = pg.gD; pc.gDist = pg.gDist;
        pc.W = tot.x; pc.M1 = tot.y; pc.M2 = tot.z;
        pc.Vtot = tot.w + 2.0f * pg.gDist * (tot.x * tot.z - tot.y * tot.y);
        pc.Tfb = T_final * ((pg.gC[0] * bg[0] + pg.gC[1] * bg[1] + pg.gC[2] * bg[2]) - pg.gA);
        pc.gMed = pg.gMed; pc.Tfin = T_final;
    }
    float P = pre.w + pc.gDist * (pc.W * pre.z - 2.0f * pc.M1 * pre.y + pc.M2 * pre.x);
    for (uint32_t w = threadIdx.x; w < kBwdChunk * (kGRec + 1); w += 256) (&sgrad[0][0])[w] = 0.0f;
    unsigned long long m[2];
    stage_segment<kBRec>(sh, c, point_list, brec, m[0], m[1]);   // (its barriers also publish the cleared gradient image)
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float pxf = c.pxf, pyf = c.pyf;
    auto pair_grad = [&](const float *b, const PairFwd &f, float test_T, float *cw) {
        const PairV v = pair_value(b, f, T, pc);
        P += v.w * v.vi;
        pair_words(f, v, T, P, pc, pxf, pyf, cw);
        T = test_T;
    };

    for (int h = 0; h < 2; ++h) {
        unsigned long long un = sh.colq[h][wv & 1] & sh.rowq[h][wv >> 1];
        un = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(un >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)un);
        if (un == 0ull) continue;
        // (pixel, entry) pairs of this wave / entries it meets: the lanes that evaluate an entry on average
        float pairs = done ? 0.0f : (float)__popcll(m[h]);
        pairs += dpp_row<0x111>(pairs); pairs += dpp_row<0x112>(pairs); pairs += dpp_row<0x114>(pairs); pairs += dpp_row<0x118>(pairs);
        const float wave_pairs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pairs), 15)) +
                                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pairs), 31)) +
                                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pairs), 47)) +
                                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pairs), 63));
        if (wave_pairs >= (float)GA_BWD_WAVE_MAJOR_LANES * (float)__popcll(un)) {
            while (un) {
                const int eh = __builtin_ctzll(un);
                un &= un - 1;
                const bool act = !done && ((m[h] >> eh) & 1ull);
                if (!__any(act)) continue;
                const int e = eh + 64 * h;
                const float *b = &sh.rec[0][e];
                PairFwd f;
                pair_forward(b, pxf, pyf, f);
                float test_T = T;
                bool on = act && f.ok;
                if (on) {
                    test_T = T * (1.0f - f.alpha);
                    if (test_T < 0.0001f) { done = true; on = false; }
                }
                float cw[kGRec];
#pragma unroll
                for (int q = 0; q < kGRec; ++q) cw[q] = 0.0f;
                if (on) pair_grad(b, f, test_T, cw);
                if (!__any(on)) continue;
                float q[5];
                wave_totals18(cw, q);
                if ((lane & 15) == 15) {   // the last lane of row r holds word 4 i + {0, 2, 1, 3}[r] of q[i]
                    const int r = lane >> 4;
                    float *g = &sgrad[e][((r & 1) << 1) | (r >> 1)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) atomicAdd(g + 4 * i, q[i]);
                    if (!(r & 1)) atomicAdd(&sgrad[e][16 + (r >> 1)], q[4]);
                }
            }
        } else {
            unsigned long long mm = done ? 0ull : m[h];
            while (mm) {
                const int e = __builtin_ctzll(mm) + 64 * h;
                mm &= mm - 1;
                const float *b = &sh.rec[0][e];
                PairFwd f;
                pair_forward(b, pxf, pyf, f);
                if (!f.ok) continue;
                const float test_T = T * (1.0f - f.alpha);
                if (test_T < 0.0001f) { done = true; break; }
                float cw[kGRec];
                pair_grad(b, f, test_T, cw);
#pragma unroll
                for (int q = 0; q < kGRec; ++q)
                    if (cw[q] != 0.0f) atomicAdd(&sgrad[e][q], cw[q]);
            }
        }
    }
    __syncthreads();   // the segment's gradient words are complete: one global atomic per word
    for (uint32_t w2 = threadIdx.x; w2 < c.cn * kGRec; w2 += 256) {
        const uint32_t e = w2 / kGRec, f2 = w2 - e * kGRec;
        const float val = sgrad[e][f2];
        if (val != 0.0f) atomicAdd(grec + (c.vbase + sh.id[e]) * kGRec + f2, val);
    }
}

#undef GA_BF

// One thread per Gaussian, the views in a loop: what is linear in the per-view gradient record -- dL/dHm (scaled axes and
// centre), the normal column of dL/dR, opacity, colour -- is summed over the views in registers, the rotation / scale /
// quaternion chain is applied once, and the five gradients leave as plain stores (no atomics, no clearing memsets, and a sum
// whose order does not depend on the schedule).
__global__ __launch_bounds__(256) void surfel_preprocess_bwd_kernel(const float *__restrict__ means3D, const float *__restrict__ scales,
                                                                    const float *__restrict__ rotations, const float *__restrict__ viewmatrix,
                                                                    const float *__restrict__ projmatrix, float scale_modifier, Dims dm,
                                                                    const int32_t *__restrict__ radii, const float *__restrict__ grec,
                                                                    float *__restrict__ d_means, float *__restrict__ d_opac,
                                                                    float *__restrict__ d_colors, float *__restrict__ d_scales,
                                                                    float *__restrict__ d_rot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dm.N) return;
    float sHm[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // sum over views of dL/dHm
    float sn[3] = {0.f, 0.f, 0.f};                                           // ... of the normal column of dL/dR
    float sop = 0.f, scol[3] = {0.f, 0.f, 0.f};
    SplatFwd s;
    bool seen = false;
    const float halfW = (float)dm.W / 2.0f, halfH = (float)dm.H / 2.0f;
    const float cW = (float)(dm.W - 1) / 2.0f, cH = (float)(dm.H - 1) / 2.0f;
    for (int v = 0; v < dm.V; ++v) {
        const size_t idx = (size_t)v * dm.N + i;
        if (radii[idx] <= 0) continue;
        seen = true;
        const float *g = grec + idx * kGRec;
        const float *vm = viewmatrix + 16 * v, *pm = projmatrix + 16 * v;
        splat_forward(means3D, scales, rotations, vm, pm, scale_modifier, dm, i, s);
        // dM[a][c]: rows a = (u axis, v axis, centre), columns c = (Tu, Tv, Tw)
        float dM[3][3];
        for (int a = 0; a < 3; ++a) { dM[a][0] = g[a]; dM[a][1] = g[3 + a]; dM[a][2] = g[6 + a]; }
        // the low-pass filter's centre: cx = sum t Tu Tw / Dn, cy = sum t Tv Tw / Dn, Dn = sum t Tw^2
        const float t[3] = {kCutoff * kCutoff, kCutoff * kCutoff, -1.0f};
        const float gcx = g[9], gcy = g[10], inv = 1.0f / s.Dn;
        for (int a = 0; a < 3; ++a) {
            dM[a][0] += gcx * t[a] * s.Tw[a] * inv;
            dM[a][1] += gcy * t[a] * s.Tw[a] * inv;
            dM[a][2] += gcx * (t[a] * s.Tu[a] * inv - s.cx * 2.0f * t[a] * s.Tw[a] * inv) +
                        gcy * (t[a] * s.Tv[a] * inv - s.cy * 2.0f * t[a] * s.Tw[a] * inv);
        }
        // M[a] = A(a) N_pix with A(a)[j'] = sum_j Hm[a][j] pm[4 j + j'] (+ pm[12 + j'] for the centre row)
        for (int a = 0; a < 3; ++a) {
            const float dA[4] = {dM[a][0] * halfW, dM[a][1] * halfH, 0.0f, dM[a][0] * cW + dM[a][1] * cH + dM[a][2]};
            for (int j = 0; j < 3; ++j) sHm[a][j] += dA[0] * pm[4 * j] + dA[1] * pm[4 * j + 1] + dA[2] * pm[4 * j + 2] + dA[3] * pm[4 * j + 3];
        }
        for (int r = 0; r < 3; ++r)   // nv_c = mult sum_r vm[4 r + c] n_r
            sn[r] += s.mult * (g[12] * vm[4 * r] + g[13] * vm[4 * r + 1] + g[14] * vm[4 * r + 2]);
        sop += g[11];
        for (int a = 0; a < 3; ++a) scol[a] += g[15 + a];
    }
    float dq_out[4] = {0.f, 0.f, 0.f, 0.f}, dsu = 0.f, dsv = 0.f;
    if (seen) {   // (s: the view-independent members -- rotation columns, scales, quaternion -- of the last view seen)
        float dR[3][3];   // dL/dR[row][col]: columns tu, tv, n
        for (int r = 0; r < 3; ++r) {
            dR[r][0] = s.su * sHm[0][r];
            dR[r][1] = s.sv * sHm[1][r];
            dsu += s.tu[r] * sHm[0][r];
            dsv += s.tv[r] * sHm[1][r];
            dR[r][2] = sn[r];
        }
        // R(q^) with q^ = (r, x, y, z) = q / |q|
        const float r = s.q[0] * s.qs, x = s.q[1] * s.qs, y = s.q[2] * s.qs, z = s.q[3] * s.qs;
        float dq[4];
        dq[0] = 2.0f * (z * (dR[1][0] - dR[0][1]) + y * (dR[0][2] - dR[2][0]) + x * (dR[2][1] - dR[1][2]));
        dq[1] = 2.0f * (y * (dR[0][1] + dR[1][0]) + z * (dR[0][2] + dR[2][0]) + r * (dR[2][1] - dR[1][2])) - 4.0f * x * (dR[1][1] + dR[2][2]);
        dq[2] = 2.0f * (x * (dR[0][1] + dR[1][0]) + r * (dR[0][2] - dR[2][0]) + z * (dR[1][2] + dR[2][1])) - 4.0f * y * (dR[0][0] + dR[2][2]);
        dq[3] = 2.0f * (r * (dR[1][0] - dR[0][1]) + x * (dR[0][2] + dR[2][0]) + y * (dR[1][2] + dR[2][1])) - 4.0f * z * (dR[0][0] + dR[1][1]);
        // q^ = q / |q|: dq = (dq^ - q^ (q^ . dq^)) / |q|
        const float dotq = r * dq[0] + x * dq[1] + y * dq[2] + z * dq[3];
        const float qh[4] = {r, x, y, z};
        for (int c = 0; c < 4; ++c) dq_out[c] = (dq[c] - qh[c] * dotq) * s.qs;
    }
    for (int a = 0; a < 3; ++a) d_means[3 * i + a] = sHm[2][a];
    d_scales[2 * i] = scale_modifier * dsu;
    d_scales[2 * i + 1] = scale_modifier * dsv;
    d_opac[i] = sop;
    for (int a = 0; a < 3; ++a) d_colors[3 * i + a] = scol[a];
    for (int c = 0; c < 4; ++c) d_rot[4 * i + c] = dq_out[c];
}

}  // namespace ga

namespace ga {

// `scratch`: brec | grec | seg_base | seg_owner | Tseg | Tend | part | totals | Tfinal | big (every section 256-byte aligned)
struct BwdScratch {
    size_t brec, grec, seg_base, seg_owner, Tseg, Tend, part, totals, Tfinal, big, total;
    uint32_t max_segs;
};

static bool bwd_scratch_layout(int64_t N, int64_t V, int64_t tiles, int64_t capacity, BwdScratch &o)
{
    if (N < 0 || V <= 0 || tiles <= 0 || capacity < 0) return false;
    const int64_t segs = capacity / kBwdSeg + V * tiles + 1;   // sum over lists of ceil(len / kBwdSeg) <= D / kBwdSeg + lists
    if (segs > 0x7fffffffll) return false;
    o.max_segs = (uint32_t)segs;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    o.brec = take((size_t)N * V * kBRec * 4);
    o.grec = take((size_t)N * V * kGRec * 4);
    o.seg_base = take((size_t)(V * tiles + 1) * 4);
    o.seg_owner = take((size_t)segs * 4);
    o.Tseg = take((size_t)segs * 256 * 4);
    o.Tend = take((size_t)segs * 256 * 4);
    o.part = take((size_t)segs * 256 * 16);
    o.totals = take((size_t)V * tiles * 256 * 16);
    o.Tfinal = take((size_t)V * tiles * 256 * 4);
    o.big = take((size_t)(2 * segs + 1) * 4);
    o.total = off;
    return true;
}

static bool bwd_dims(const GaSurfelForwardArgs &f, Dims &d)
{
    const int64_t gx = ((int64_t)f.image_width + kTile - 1) / kTile, gy = ((int64_t)f.image_height + kTile - 1) / kTile;
    if (f.num_points < 0 || f.num_views <= 0 || f.image_height <= 0 || f.image_width <= 0) return false;
    d.N = f.num_points; d.V = f.num_views; d.H = f.image_height; d.W = f.image_width; d.gx = (int)gx; d.gy = (int)gy; d.tiles = (int)(gx * gy);
    return true;
}

}  // namespace ga

extern "C" int ga_surfel_backward(const GaSurfelBackwardArgs *a, void *stream_v)
{
    using namespace ga;
    if (!a) return GA_ERR_NULL_ARG;
    const GaSurfelForwardArgs &f = a->fwd;
    Dims d;
    if (!bwd_dims(f, d)) return GA_ERR_BAD_SHAPE;
    GaSurfelWorkspaceLayout L;
    const int rc = ga_surfel_workspace_layout2(d.N, d.V, d.H, d.W, f.capacity, f.seg_capacity, &L);   // (the forward's own layout)
    if (rc != GA_OK) return rc;
    if (f.workspace && f.workspace_bytes < L.total_bytes) return GA_ERR_WORKSPACE;
    if (!f.workspace || !f.out_color || !f.out_others || !f.radii || !f.bg || !f.viewmatrix ||
        !f.projmatrix || !a->grad_color || !a->grad_others || !a->scratch || !a->grad_means3D || !a->grad_opacities ||
        !a->grad_colors || !a->grad_scales || !a->grad_rotations)
        return GA_ERR_NULL_ARG;
    if (d.N > 0 && (!f.means3D || !f.opacities || !f.colors || !f.scales || !f.rotations)) return GA_ERR_NULL_ARG;
    BwdScratch sc;
    if (!bwd_scratch_layout(d.N, d.V, d.tiles, f.capacity, sc)) return GA_ERR_BAD_SHAPE;
    if (a->scratch_bytes < sc.total || (reinterpret_cast<uintptr_t>(a->scratch) & 15)) return GA_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_v);
    (void)hipGetLastError();
    if (d.N == 0) return GA_OK;
    unsigned char *w = static_cast<unsigned char *>(f.workspace);
    const int64_t *status = reinterpret_cast<const int64_t *>(w + L.status);
    const uint32_t *tile_start = reinterpret_cast<const uint32_t *>(w + L.tile_start);
    const uint32_t *point_list = reinterpret_cast<const uint32_t *>(w + L.point_list);
    unsigned char *sp = static_cast<unsigned char *>(a->scratch);
    float *brec = reinterpret_cast<float *>(sp + sc.brec);
    float *grec = reinterpret_cast<float *>(sp + sc.grec);
    BwdPlan pl;
    pl.seg_base = reinterpret_cast<uint32_t *>(sp + sc.seg_base);
    pl.seg_owner = reinterpret_cast<uint32_t *>(sp + sc.seg_owner);
    // the transmittance every pixel enters every segment with: left by the forward (fwd.seg_T), or formed here by two launches
    const bool have_T = f.seg_T != nullptr;
    if (have_T && f.seg_T_floats < (int64_t)sc.max_segs * 256) return GA_ERR_WORKSPACE;
    pl.Tseg = have_T ? f.seg_T : reinterpret_cast<float *>(sp + sc.Tseg);
    pl.Tend = reinterpret_cast<float *>(sp + sc.Tend);
    pl.part = reinterpret_cast<float4 *>(sp + sc.part);
    pl.total = reinterpret_cast<float4 *>(sp + sc.totals);
    pl.Tfinal = reinterpret_cast<float *>(sp + sc.Tfinal);
    pl.big = reinterpret_cast<uint32_t *>(sp + sc.big);
    pl.max_segs = sc.max_segs;
    const dim3 gridN((unsigned)((d.N + 255) / 256), (unsigned)d.V);
    hipLaunchKernelGGL(surfel_bwd_record_kernel, dim3((unsigned)((d.N + 1023) / 1024), (unsigned)d.V + 1u), dim3(1024), 0, s, f.means3D,
                       f.opacities, f.colors, f.scales, f.rotations, f.viewmatrix, f.projmatrix, f.scale_modifier, d, f.radii,
                       reinterpret_cast<const float *>(w + L.record), brec, grec, tile_start, pl, status);
    // (the grid covers the bound on the number of segments; the workgroups past the real count leave at once)
    const dim3 gridS(sc.max_segs);
    if (!have_T) {
        hipLaunchKernelGGL(surfel_bwd_trans_kernel, gridS, dim3(256), 0, s, tile_start, point_list, brec, d, pl);
        hipLaunchKernelGGL(surfel_bwd_prefix_T_kernel, dim3((unsigned)(d.V * d.tiles)), dim3(256), 0, s, tile_start, pl, d.V * d.tiles);
    }
    hipLaunchKernelGGL(surfel_bwd_sums_kernel, gridS, dim3(256), 0, s, tile_start, point_list, brec, d, pl, a->grad_color,
                       a->grad_others);
    hipLaunchKernelGGL(surfel_bwd_prefix_sums_kernel, dim3((unsigned)(d.V * d.tiles)), dim3(256), 0, s, tile_start, pl, d.V * d.tiles);
    hipLaunchKernelGGL(surfel_bwd_grad_kernel, gridS, dim3(256), 0, s, tile_start, point_list, brec, f.bg, d, pl,
                       a->grad_color, a->grad_others, grec);
    hipLaunchKernelGGL(surfel_bwd_grad_big_kernel, gridS, dim3(256), 0, s, tile_start, point_list, brec, f.bg, d, pl,
                       a->grad_color, a->grad_others, grec);
    hipLaunchKernelGGL(surfel_preprocess_bwd_kernel, dim3(gridN.x), dim3(256), 0, s, f.means3D, f.scales, f.rotations, f.viewmatrix,
                       f.projmatrix, f.scale_modifier, d, f.radii, grec, a->grad_means3D, a->grad_opacities, a->grad_colors,
                       a->grad_scales, a->grad_rotations);
    return hipGetLastError() == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

extern "C" size_t ga_surfel_backward_scratch_bytes(const GaSurfelForwardArgs *fwd)
{
    ga::Dims d;
    ga::BwdScratch sc;
    if (!fwd || !ga::bwd_dims(*fwd, d) || !ga::bwd_scratch_layout(d.N, d.V, d.tiles, fwd->capacity, sc)) return 0;
    return sc.total;
}
