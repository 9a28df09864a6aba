// surfel_blend.hip -- per-tile front-to-back alpha compositing of colour + depth + normal (+ median depth,
// distortion), gfx950.  Replaces upstream renderCUDA of diff_surfel_rasterization (call site
// /root/reference/nsr/gs_surfel.py:100-114; consumer of the 7 allmap channels :121-142); arithmetic per
// SURVEY.md Appendix A.1 "Blend".
//
// MI355X-first formulation ("wave-autonomous" blend), not the CUDA block-cooperative one.  What the measurements on
// MI355X showed (profiles/r1a_*): the kernel is bound by the SERIAL CHAIN of the longest tile lists and by VALU issue,
// never by HBM; so the design minimises latency on the chain and instructions per (pixel, splat) pair:
//   * a 256-thread workgroup still owns one 16x16 tile (that granularity is part of the semantics: the tile rect
//     decides which pixels a splat may touch), but each of its four 64-lane wavefronts owns one 8x8 quadrant and runs
//     completely on its own -- no workgroup barrier, independent early termination; workgroups are scheduled longest
//     list first (tile_order);
//   * the tile's depth-ordered list is consumed 64 entries at a time with LANES = ENTRIES: every lane fetches one
//     entry's conservative {alpha >= 1/255} pixel box and its 96-byte record (vector loads, all 64 in flight at
//     once, issued one chunk AHEAD of use), tests the box against the quadrant, rebases the record's plane
//     coefficients to the quadrant origin and parks it in a wave-private LDS slot.  A 64-bit ballot of the box test
//     is the list of entries that can contribute (about half are culled for the sub-pixel splats of real scenes,
//     which also halves the serial chain);
//   * LANES = PIXELS with per-lane survivor lists: from 16 ballots per chunk (does my entry's box cover pixel column c /
//     row r of the quadrant?) every lane ANDs the masks of its own column and row and walks only those entries --
//     the trip count of a wave drops from "survivors of the quadrant" to "survivors of its busiest pixel"; records are
//     gathered from LDS four at a time, their alphas evaluated back to back (no cross-entry dependence), then
//     composited in order;
//   * the ray/splat intersection uses the plane form p = C' + dx*A + dy*B (6 FMAs) instead of two 3-vector affine
//     maps and a cross product (18 ops); A, B, C come from the preprocess kernel;
//   * upstream's chain of `continue` filters is evaluated branch-free into one predicate, so a pair costs ~25 VALU
//     instructions and one EXEC-masked region (~25 more) when it contributes; a wave leaves the list as soon as its
//     64 pixels are saturated.
// Pixel results are compared with the oracle by MSE (<= 1e-5, tests/), so this TU may contract to FMA and uses
// v_rcp_f32 / v_exp_f32 instead of IEEE division and libm expf.
#include <hip/hip_fp16.h>

#include "surfel_common.h"

namespace ga {

typedef float f2 __attribute__((ext_vector_type(2)));  // one packed-fp32 operand (even-aligned register pair)

__device__ __forceinline__ f2 lo2(const float4 &q) { return f2{q.x, q.y}; }
__device__ __forceinline__ f2 hi2(const float4 &q) { return f2{q.z, q.w}; }

struct PixelAcc {  // pairs are updated by one v_pk_fma_f32
    float T, Dp, dist, median;
    f2 M;     // M1, M2
    f2 N01;   // normal x, y
    f2 N2C0;  // normal z, red
    f2 C12;   // green, blue
};

__device__ __forceinline__ PixelAcc fresh_pixel(float T)
{
    return PixelAcc{T, 0.0f, 0.0f, 0.0f, f2{0.0f, 0.0f}, f2{0.0f, 0.0f}, f2{0.0f, 0.0f}, f2{0.0f, 0.0f}};
}

struct Rec {  // one staged record (quadrant-relative, see the staging step) in registers; layout: surfel_common.h
    float4 q0, q1, q2, q3, q4;
    f2 q5;
};

// LDS image of a wave's staged chunks: six planes of 128 float4 (plane q holds quad q of every slot), so that the
// staging writes are contiguous and a gather of 16 different slots spreads over all 64 banks.
__device__ __forceinline__ Rec lds_read_rec(const float4 (*planes)[128], int j)
{
    return Rec{planes[0][j], planes[1][j], planes[2][j], planes[3][j], planes[4][j],
               *reinterpret_cast<const f2 *>(&planes[5][j])};
}

// One (pixel, splat) evaluation -- SURVEY.md A.1 "Blend" -- in two parts.  dxy: this lane's pixel relative to the
// quadrant origin.  eval_alpha is free of cross-entry dependences (several entries are evaluated back to back);
// composite is the short sequential part.  Upstream's chain of `continue` filters is evaluated branch-free into one
// predicate (the filters commute: each one only decides whether the pair is skipped).
struct Alpha {
    float alpha;
    f2 s;
    bool pass, use3d;
};

__device__ __forceinline__ Alpha eval_alpha(const Rec &r, f2 dxy)
{
    // p = C' + dx*A + dy*B
    const f2 pxy = dxy.y * hi2(r.q0) + (dxy.x * lo2(r.q0) + lo2(r.q1));
    const float p2 = fmaf(dxy.y, r.q1.w, fmaf(dxy.x, r.q1.z, r.q2.z));
    const float rz = __builtin_amdgcn_rcpf(p2);
    Alpha o;
    o.s = pxy * rz;
    const f2 ss = o.s * o.s;
    const float rho3d = ss.x + ss.y;
    const f2 e = lo2(r.q2) - dxy;  // centre - pixel
    const f2 ee = e * e;
    const float rho2d = kFilterInvSquare * (ee.x + ee.y);
    const float rho = fminf(rho3d, rho2d);
    o.use3d = rho3d <= rho2d;
    o.alpha = fminf(0.99f, r.q2.w * __builtin_amdgcn_exp2f(rho * -0.72134752044f));
    // upstream: p.z == 0 -> skip ; power = -0.5*rho > 0 -> skip (a NaN rho passes) ; alpha < 1/255 -> skip
    o.pass = p2 != 0.0f && !(rho < 0.0f) && !(o.alpha < 1.0f / 255.0f);
    return o;
}

__device__ __forceinline__ float pair_depth(const Rec &r, const Alpha &e)
{
    const f2 d = e.s * lo2(r.q3);
    return e.use3d ? (d.x + d.y) + r.q3.z : r.q3.z;
}

__device__ __forceinline__ void composite(const Rec &r, const Alpha &e, PixelAcc &a, bool &done)
{
    const float kM = kFar / (kFar - kNear);
    const float depth = pair_depth(r, e);
    const float test_T = a.T * (1.0f - e.alpha);
    const bool near_ok = !(depth < kNear);            // upstream: depth < near -> skip (before the alpha test)
    const bool stop = near_ok && test_T < 0.0001f;    // upstream: done = true
    done = done || stop;
    if (near_ok && !stop) {
        const float w = e.alpha * a.T;
        const float A = 1.0f - a.T;
        const float m = kM * (1.0f - kNear * __builtin_amdgcn_rcpf(depth));
        const f2 mm = f2{m, m * m};
        a.dist += (mm.y * A + a.M.y - 2.0f * m * a.M.x) * w;
        a.Dp += depth * w;
        a.M += mm * w;
        if (a.T > 0.5f) a.median = depth;
        a.N01 += lo2(r.q4) * w;
        a.N2C0 += hi2(r.q4) * w;
        a.C12 += r.q5 * w;
        a.T = test_T;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Walk the list range [sbeg, send) for one wave = one 8x8 pixel quadrant.  FULL: the complete per-pixel blend;
// !FULL: only the transmittance product of the range (every contributing pair multiplies T by 1 - alpha; no stop
// rule), which the segment-parallel kernel needs to give each segment its true starting transmittance.
struct WaveCtx {
    int lane;
    f2 dxy;
    float qxlo, qylo;
    uint32_t safe;  // a valid list position: out-of-range lanes re-read it
    const uint32_t *__restrict__ point_list;
    const float4 *__restrict__ rec4;
    float4 (*planes)[128];
};

template <bool FULL>
__device__ __forceinline__ void walk_list(const WaveCtx &c, uint32_t sbeg, uint32_t send, PixelAcc &a, bool &done,
                                          unsigned &stat_iters, unsigned &stat_chunks, int flags, unsigned &stat_useful)
{
    if (sbeg >= send) return;
    const int lane = c.lane;
    float4(*planes)[128] = c.planes;
    // ---- software pipeline over 64-entry chunks (lanes = entries) ------------------------------------------------
    //   iteration k consumes {bb, g0..g5} of chunk k (issued during k-1), issues them for chunk k+1 (whose ids were
    //   issued during k-1) and issues the ids of chunk k+2.  Loads are unconditional (out-of-range lanes re-read a
    //   valid entry) so the loop body is straight-line code and the loaded registers stay untouched until consumed.
    float4 g0, g1, g2, g3, g4;
    f2 g5;
    uint32_t id_next;
    {
        const uint32_t e0 = sbeg + lane < send ? sbeg + lane : c.safe;
        const uint32_t e1 = sbeg + 64 + lane < send ? sbeg + 64 + lane : c.safe;
        const uint32_t id0 = c.point_list[e0];
        id_next = c.point_list[e1];
        const float4 *r = c.rec4 + (size_t)id0 * 6;
        g0 = r[0]; g1 = r[1]; g2 = r[2]; g3 = r[3]; g4 = r[4]; g5 = *reinterpret_cast<const f2 *>(r + 5);
    }
    // loop-invariant lane predicates "my column / row is c" as wave masks (SGPR pairs)
    unsigned long long colsel[8], rowsel[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        colsel[k] = __builtin_amdgcn_ballot_w64((lane & 7) == k);
        rowsel[k] = __builtin_amdgcn_ballot_w64((lane >> 3) == k);
    }

    // Per-lane survivor masks of a TWO-chunk window: `cur` = what is left of the previous chunk (LDS buffer oldb), `nxt` =
    // the chunk staged in this step (buffer newb).  A lane that has finished `cur` runs ahead into `nxt` while slower
    // lanes still work on `cur`; the step ends when no lane has anything left in `cur`, which frees that buffer for the
    // chunk after next.  With one-chunk windows only 36 % of the lane slots did work (lists per chunk are short and
    // Poisson-like: mean 3.4, max over 64 lanes ~9); the run-ahead evens that out (measured: see DESIGN.md).
    unsigned long long cur = 0, nxt = 0;
    int step = 0;
    constexpr int kU = 4;
    auto trips = [&](int oldb, int newb) {
        // Lanes walk their own lists independently (compositing order only matters per pixel), kU entries per trip: the
        // kU gathers and alpha evaluations are mutually independent, then the contributing ones are composited in
        // order.  An exhausted lane re-reads its last slot with the live flag off.
        while (__builtin_amdgcn_ballot_w64(cur != 0) != 0) {
            int j[kU];
            bool live[kU];
            int last = lane;
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const bool has = cur != 0;
                const unsigned long long sel = has ? cur : nxt;
                live[u] = sel != 0;
                j[u] = live[u] ? __builtin_ctzll(sel) + (has ? oldb : newb) : last;
                last = j[u];
                const unsigned long long rest = sel & (sel - 1);
                cur = has ? rest : 0ull;
                nxt = has ? nxt : rest;
            }
            stat_iters += kU;
            Rec r[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (FULL) r[u] = lds_read_rec(planes, j[u]);
                else { r[u].q0 = planes[0][j[u]]; r[u].q1 = planes[1][j[u]]; r[u].q2 = planes[2][j[u]]; r[u].q3 = planes[3][j[u]]; }
            }
            Alpha e[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) e[u] = eval_alpha(r[u], c.dxy);
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (FULL) {
                    if (live[u] && e[u].pass && !done) composite(r[u], e[u], a, done);
                } else {
                    const float depth = pair_depth(r[u], e[u]);
                    if (live[u] && e[u].pass && !done && !(depth < kNear)) {
                        // T_global <= T_segment: once the segment product alone trips the stop rule the sequential
                        // loop has stopped at or before this pair, and every later segment is dead (P = 0)
                        const float t = a.T * (1.0f - e[u].alpha);
                        done = t < 0.0001f;
                        a.T = done ? 0.0f : t;
                    }
                }
            }
            if (done) { cur = 0; nxt = 0; }
        }
    };

    for (uint32_t base = sbeg; base < send; base += 64, ++step) {
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
        const int newb = (step & 1) * 64, oldb = 64 - newb;
        // ---- lanes = entries: which pixel columns / rows of the quadrant does my entry's cull box cover? ----------
        const bool valid = base + lane < send;
        // centre relative to the quadrant origin and the cull half-extents (fp16 pair, +inf = unbounded, < 0 = never)
        const float ex0 = g2.x - c.qxlo, ey0 = g2.y - c.qylo;
        const uint32_t cull = __float_as_uint(g3.w);
        const float rx = __half2float(__ushort_as_half((unsigned short)(cull & 0xffffu)));
        const float ry = __half2float(__ushort_as_half((unsigned short)(cull >> 16)));
        unsigned long long xm[8], ym[8], xany = 0, yany = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            xm[k] = __builtin_amdgcn_ballot_w64(valid && fabsf((float)k - ex0) <= rx);
            ym[k] = __builtin_amdgcn_ballot_w64(valid && fabsf((float)k - ey0) <= ry);
            xany |= xm[k];
            yany |= ym[k];
        }
        const unsigned long long hitmask = xany & yany;
        if ((hitmask >> lane) & 1ull) {
            // rebase to the quadrant origin: C' = C + (q0.x - ox)*A + (q0.y - oy)*B with o = rint(centre); centre -= q0
            const float ux = c.qxlo - rintf(g2.x), uy = c.qylo - rintf(g2.y);
            const float Cx = fmaf(uy, g0.z, fmaf(ux, g0.x, g1.x));
            const float Cy = fmaf(uy, g0.w, fmaf(ux, g0.y, g1.y));
            const float Cz = fmaf(uy, g1.w, fmaf(ux, g1.z, g2.z));
            const int slot = newb + lane;
            planes[0][slot] = g0;
            planes[1][slot] = make_float4(Cx, Cy, g1.z, g1.w);
            planes[2][slot] = make_float4(ex0, ey0, Cz, g2.w);
            planes[3][slot] = make_float4(g3.x, g3.y, g3.z, g3.w);
            if (FULL) {
                planes[4][slot] = g4;
                *reinterpret_cast<f2 *>(&planes[5][slot]) = g5;
            }
        }
        {   // issue the next chunk's loads (ids arrived during the previous iteration) and the ids after that
            const uint32_t idn = id_next;
            const uint32_t e2 = base + 128 + lane < send ? base + 128 + lane : c.safe;
            id_next = c.point_list[e2];
            const float4 *r = c.rec4 + (size_t)idn * 6;
            g0 = r[0]; g1 = r[1]; g2 = r[2]; g3 = r[3];
            if (FULL) { g4 = r[4]; g5 = *reinterpret_cast<const f2 *>(r + 5); }
        }
        ++stat_chunks;
        if (flags & 2) continue;  // flag 2: staging only (measurement aid, not in the public header)
        // ---- lanes = pixels: my own survivor list = entries whose box covers MY column and MY row ---------------
        unsigned long long mx = xm[0], my = ym[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            mx = ((colsel[k] >> lane) & 1ull) ? xm[k] : mx;
            my = ((rowsel[k] >> lane) & 1ull) ? ym[k] : my;
        }
        nxt = done ? 0ull : (mx & my);
        if (flags & GA_SURFEL_FLAG_STATS) {
            unsigned pc = __builtin_popcountll(nxt);
            for (int o = 32; o > 0; o >>= 1) pc += __shfl_xor(pc, o, 64);
            stat_useful += pc;
        }
        trips(oldb, newb);   // until every lane has finished the previous chunk
        cur = nxt;           // what is left of this chunk becomes the "previous chunk" of the next step
        nxt = 0;
    }
    trips(64 - (step & 1) * 64, 0);  // drain: `cur` is the last staged chunk (buffer of step-1), `nxt` is empty
}

__device__ __forceinline__ void write_pixel(const PixelAcc &a, const float *__restrict__ bg, const Dims &dm, int v, int pxi,
                                            int pyi, float *__restrict__ out_color, float *__restrict__ out_others)
{
    const size_t HW = (size_t)dm.H * dm.W, pid = (size_t)pyi * dm.W + pxi;
    float *oc = out_color + (size_t)v * 3 * HW + pid;
    float *oo = out_others + (size_t)v * 7 * HW + pid;
    oc[0] = a.N2C0.y + a.T * bg[0];
    oc[HW] = a.C12.x + a.T * bg[1];
    oc[2 * HW] = a.C12.y + a.T * bg[2];
    oo[0] = a.Dp;
    oo[HW] = 1.0f - a.T;
    oo[2 * HW] = a.N01.x;
    oo[3 * HW] = a.N01.y;
    oo[4 * HW] = a.N2C0.x;
    oo[5 * HW] = a.median;
    oo[6 * HW] = a.dist;
}

// ---------------------------------------------------------------------------------------------------------------
// Blend kernel.  The serial per-pixel chain of the longest lists was the critical path of the whole rasterizer (PMC:
// ~1 resident wave per SIMD, VALU 20 % busy), so lists are handled in two shapes:
//   * short list (< kLongList pairs): one workgroup per 16x16 tile, each wave blends one 8x8 quadrant over the whole list;
//   * long list: one workgroup per 8x8 QUADRANT whose four waves take contiguous quarters of the list (SEGMENTS):
//       pass 1  every wave multiplies (1 - alpha) over the contributing pairs of its segment -> seg_T[w][pixel];
//               P_w = prod_{j<w} seg_T[j] is the transmittance the sequential algorithm has on entering segment w;
//       pass 2  every wave runs the UNCHANGED sequential blend over its segment starting from T = P_w: weights, the
//               `T > 0.5` median test and the `T (1-alpha) < 1e-4` stop rule therefore see the global transmittance;
//       merge   wave 0 adds the segments in order; the depth-distortion prefix sums M1, M2 are segment-local, their
//               cross terms  M2_before * W_k - 2 M1_before * M1_k  (W_k = the segment's summed weights = P_k - T_end,k)
//               are added here; a segment after one that hit the stop rule contributes nothing, exactly as the
//               sequential loop would have left it.
//     The only numerical difference to the sequential order is the rounding of P_w (a product of segment products).
// The tile scan leaves the long tiles at the front of tile_order and their number in status[GA_STATUS_LONG_TILES]; the
// grid is [4 x long tiles rounded up to 8 | remaining tiles], sized on the host from the capacity bound.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void surfel_blend_kernel(const uint4 *__restrict__ tile_order,
                                                               const uint32_t *__restrict__ point_list,
                                                               const float *__restrict__ record,
                                                               const float *__restrict__ bg, Dims dm, int ntiles,
                                                               float *__restrict__ out_color,
                                                               float *__restrict__ out_others,
                                                               int64_t *__restrict__ status, int flags)
{
    __shared__ __attribute__((aligned(16))) float4 stage[4][6][128];  // wave-private record planes, 24 KiB; after pass 2
                                                                      // the same 6 KiB hold the wave's segment results
    __shared__ float seg_T[4][64];                                    // pass-1 transmittance of each segment
    const int64_t overflow = status[GA_STATUS_OVERFLOW], nlong64 = status[GA_STATUS_LONG_TILES];  // one latency
    if (overflow) return;
    const uint32_t nlong = (uint32_t)nlong64, nlong8 = (nlong + 7u) & ~7u;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t pos, quad;
    bool split;
    if (blockIdx.x < 4 * nlong8) {
        // blocks b, b+8, b+16, b+24 (same XCD under round-robin dispatch) are the four quadrants of one long tile
        const uint32_t grp = blockIdx.x >> 5, xcd = blockIdx.x & 7;
        quad = (blockIdx.x >> 3) & 3;
        pos = grp * 8 + xcd;
        if (pos >= nlong) return;
        split = true;
    } else {
        pos = nlong + (blockIdx.x - 4 * nlong8);
        if (pos >= (uint32_t)ntiles) return;
        quad = (uint32_t)wave;
        split = false;
    }
    const uint4 sched = tile_order[pos];  // longest lists first: (tile, list begin, list length)
    const uint32_t vt = sched.x;
    const int v = (int)(vt / (uint32_t)dm.tiles), tile = (int)(vt - (uint32_t)v * dm.tiles);
    const int tx = tile % dm.gx, ty = tile / dm.gx;
    const int qx0 = tx * kTile + (int)(quad & 1) * 8, qy0 = ty * kTile + (int)(quad >> 1) * 8;
    if (qx0 >= dm.W || qy0 >= dm.H) return;  // uniform per wave (short) or per workgroup (long): barriers stay safe
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < dm.W && pyi < dm.H;

    const uint32_t beg = sched.y, end = sched.y + sched.z;
    const size_t vbase = (size_t)v * dm.N;
    WaveCtx c;
    c.lane = lane; c.dxy = f2{(float)(lane & 7), (float)(lane >> 3)}; c.qxlo = (float)qx0; c.qylo = (float)qy0;
    c.safe = beg; c.point_list = point_list;
    c.rec4 = reinterpret_cast<const float4 *>(record) + vbase * (kRec / 4);
    c.planes = stage[wave];
    unsigned stat_iters = 0, stat_chunks = 0, stat_useful = 0;
    PixelAcc a = fresh_pixel(1.0f);
    bool done = !inside;

    if (!split) {
        walk_list<true>(c, beg, end, a, done, stat_iters, stat_chunks, flags, stat_useful);
        if (inside) write_pixel(a, bg, dm, v, pxi, pyi, out_color, out_others);
    } else {
        constexpr int kSeg = 4;
        const uint32_t n = end - beg, chunks = (n + 63) / 64, cps = (chunks + kSeg - 1) / kSeg;
        const uint32_t sbeg = min(end, beg + (uint32_t)wave * cps * 64), send = min(end, sbeg + cps * 64);
        {   // pass 1: transmittance of my segment
            bool d1 = !inside;
            walk_list<false>(c, sbeg, send, a, d1, stat_iters, stat_chunks, flags, stat_useful);
        }
        seg_T[wave][lane] = a.T;
        __syncthreads();
        float P = 1.0f;
        for (int k = 0; k < wave; ++k) P *= seg_T[k][lane];
        // pass 2: the sequential blend of my segment, entered with the global transmittance
        a = fresh_pixel(P);
        done = done || P < 0.0001f;  // T never falls below 1e-4 in the sequential loop: it stopped before this segment
        a.median = -1.0f;  // depths are >= near > 0: a negative median means "not set inside this segment"
        walk_list<true>(c, sbeg, send, a, done, stat_iters, stat_chunks, flags, stat_useful);
        float *o = reinterpret_cast<float *>(stage[wave]) + lane;
        o[0 * 64] = a.N2C0.y; o[1 * 64] = a.C12.x; o[2 * 64] = a.C12.y; o[3 * 64] = a.N01.x; o[4 * 64] = a.N01.y;
        o[5 * 64] = a.N2C0.x; o[6 * 64] = a.Dp; o[7 * 64] = a.M.x; o[8 * 64] = a.M.y; o[9 * 64] = a.dist;
        o[10 * 64] = a.median;
        o[11 * 64] = a.T; o[12 * 64] = done ? 1.0f : 0.0f; o[13 * 64] = P;
        __syncthreads();
        if (wave == 0 && inside) {
            PixelAcc r = fresh_pixel(1.0f);
            bool dead = false;
#pragma unroll
            for (int k = 0; k < kSeg; ++k) {
                const float *q = reinterpret_cast<const float *>(stage[k]) + lane;
                if (!dead) {
                    const float Wk = q[13 * 64] - q[11 * 64];
                    r.N2C0.y += q[0 * 64]; r.C12.x += q[1 * 64]; r.C12.y += q[2 * 64];
                    r.N01.x += q[3 * 64]; r.N01.y += q[4 * 64]; r.N2C0.x += q[5 * 64];
                    r.Dp += q[6 * 64];
                    r.dist += q[9 * 64] + r.M.y * Wk - 2.0f * r.M.x * q[7 * 64];
                    r.M.x += q[7 * 64];
                    r.M.y += q[8 * 64];
                    if (q[10 * 64] >= 0.0f) r.median = q[10 * 64];
                    r.T = q[11 * 64];
                    dead = q[12 * 64] != 0.0f;
                }
            }
            write_pixel(r, bg, dm, v, pxi, pyi, out_color, out_others);
        }
    }
    if ((flags & GA_SURFEL_FLAG_STATS) && lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_ITERS), (unsigned long long)stat_iters);
        atomicMax(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_MAX_ITERS), (unsigned long long)stat_iters);
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_LANE_SLOTS), (unsigned long long)stat_useful);
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_CHUNKS), (unsigned long long)stat_chunks);
    }
}


void launch_blend(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s)
{
    const int nt = d.V * d.tiles;
    // long tiles hold >= long_list() pairs each, so there are at most capacity / long_list() of them
    const int64_t max_long = std::min<int64_t>(nt, a.capacity / long_list());
    const unsigned grid = (unsigned)(4 * ((max_long + 7) / 8 * 8) + nt);
    hipLaunchKernelGGL(surfel_blend_kernel, dim3(grid), dim3(256), 0, s,
                       ws.tile_order, ws.point_list, ws.record, a.bg, d, nt, a.out_color, a.out_others, ws.status,
                       a.flags);
}

}  // namespace ga
