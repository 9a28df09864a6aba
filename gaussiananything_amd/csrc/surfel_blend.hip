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
#include "surfel_common.h"

namespace ga {

struct PixelAcc {
    float T, C0, C1, C2, N0, N1, N2, Dp, M1, M2, dist, median;
};

struct Rec {  // one staged record (quadrant-relative, see the staging step) in registers
    float4 q0, q1, q2, q3, q4;
    float cb;
};

// LDS image of a wave's staged chunk: six planes of 64 float4 (plane q holds quad q of every slot), so that the
// staging writes are contiguous and a gather of 16 different slots spreads over all 64 banks.
__device__ __forceinline__ Rec lds_read_rec(const float4 (*planes)[64], int j)
{
    return Rec{planes[0][j], planes[1][j], planes[2][j], planes[3][j], planes[4][j], planes[5][j].x};
}

// One (pixel, splat) evaluation -- SURVEY.md A.1 "Blend" -- in two parts.  dx, dy: this lane's pixel relative to the
// quadrant origin.  eval_alpha is free of cross-entry dependences (several entries are evaluated back to back in one
// basic block, which lets a lone wave issue at ~2.4 instead of ~4.4 cycles per instruction); composite is the short
// sequential part.  Upstream's chain of `continue` filters is evaluated branch-free into one predicate (the filters
// commute: each one only decides whether the pair is skipped).
struct Alpha {
    float alpha, sx, sy;
    bool pass, use3d;
};

__device__ __forceinline__ Alpha eval_alpha(const Rec &r, float dx, float dy)
{
    // p = C' + dx*A + dy*B
    const float p0 = fmaf(dy, r.q0.w, fmaf(dx, r.q0.x, r.q1.z));
    const float p1 = fmaf(dy, r.q1.x, fmaf(dx, r.q0.y, r.q1.w));
    const float p2 = fmaf(dy, r.q1.y, fmaf(dx, r.q0.z, r.q2.x));
    const float rz = __builtin_amdgcn_rcpf(p2);
    Alpha o;
    o.sx = p0 * rz;
    o.sy = p1 * rz;
    const float rho3d = o.sx * o.sx + o.sy * o.sy;
    const float ex = r.q2.y - dx, ey = r.q2.z - dy;  // centre - pixel
    const float rho2d = kFilterInvSquare * (ex * ex + ey * ey);
    const float rho = fminf(rho3d, rho2d);
    o.use3d = rho3d <= rho2d;
    o.alpha = fminf(0.99f, r.q2.w * __builtin_amdgcn_exp2f(rho * -0.72134752044f));
    // upstream: p.z == 0 -> skip ; power = -0.5*rho > 0 -> skip (a NaN rho passes) ; alpha < 1/255 -> skip
    o.pass = p2 != 0.0f && !(rho < 0.0f) && !(o.alpha < 1.0f / 255.0f);
    return o;
}

__device__ __forceinline__ void composite(const Rec &r, const Alpha &e, PixelAcc &a, bool &done)
{
    const float kM = kFar / (kFar - kNear);
    const float depth = e.use3d ? fmaf(e.sx, r.q3.x, e.sy * r.q3.y) + r.q3.z : r.q3.z;
    const float test_T = a.T * (1.0f - e.alpha);
    const bool near_ok = !(depth < kNear);            // upstream: depth < near -> skip (before the alpha test)
    const bool stop = near_ok && test_T < 0.0001f;    // upstream: done = true
    done = done || stop;
    if (near_ok && !stop) {
        const float w = e.alpha * a.T;
        const float A = 1.0f - a.T;
        const float m = kM * (1.0f - kNear * __builtin_amdgcn_rcpf(depth));
        a.dist += (m * m * A + a.M2 - 2.0f * m * a.M1) * w;
        a.Dp += depth * w;
        a.M1 += m * w;
        a.M2 += m * m * w;
        if (a.T > 0.5f) a.median = depth;
        a.N0 += r.q3.w * w; a.N1 += r.q4.x * w; a.N2 += r.q4.y * w;
        a.C0 += r.q4.z * w; a.C1 += r.q4.w * w; a.C2 += r.cb * w;
        a.T = test_T;
    }
}

__global__ __launch_bounds__(256) void surfel_blend_kernel(const uint32_t *__restrict__ tile_start,
                                                           const uint32_t *__restrict__ tile_order,
                                                           const uint32_t *__restrict__ point_list,
                                                           const float *__restrict__ bbox,
                                                           const float *__restrict__ record,
                                                           const float *__restrict__ bg, Dims dm,
                                                           float *__restrict__ out_color,
                                                           float *__restrict__ out_others,
                                                           int64_t *__restrict__ status, int flags)
{
    __shared__ __attribute__((aligned(16))) float4 stage[4][6][64];  // wave-private record planes, 24 KiB
    if (status[GA_STATUS_OVERFLOW]) return;
    const uint32_t vt = tile_order[blockIdx.x];  // longest lists first (surfel_tile_scan_kernel)
    const int v = (int)(vt / (uint32_t)dm.tiles), tile = (int)(vt - (uint32_t)v * dm.tiles);
    const int tx = tile % dm.gx, ty = tile / dm.gx;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qx0 = tx * kTile + (wave & 1) * 8, qy0 = ty * kTile + (wave >> 1) * 8;
    if (qx0 >= dm.W || qy0 >= dm.H) return;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < dm.W && pyi < dm.H;
    const float dx = (float)(lane & 7), dy = (float)(lane >> 3);
    const float qxlo = (float)qx0, qylo = (float)qy0;

    const uint32_t beg = tile_start[vt], end = tile_start[vt + 1];
    const size_t vbase = (size_t)v * dm.N;
    const float4 *__restrict__ bbox4 = reinterpret_cast<const float4 *>(bbox) + vbase;
    const float4 *__restrict__ rec4 = reinterpret_cast<const float4 *>(record) + vbase * (kRec / 4);
    float4(*planes)[64] = stage[wave];

    PixelAcc a = {1.0f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned stat_iters = 0, stat_chunks = 0;
    bool done = !inside;

    // ---- software pipeline over 64-entry chunks (lanes = entries) ------------------------------------------------
    //   iteration k consumes {bb, g0..g5} of chunk k (issued during k-1), issues them for chunk k+1 (whose ids were
    //   issued during k-1) and issues the ids of chunk k+2
    // Loads are unconditional (out-of-range lanes re-read entry `beg`, always valid when the list is non-empty) so that
    // the loop body is straight-line code and the loaded registers stay untouched until the next iteration.
    float4 bb, g0, g1, g2, g3, g4, g5;
    uint32_t id_next;
    {
        const uint32_t e0 = beg + lane < end ? beg + lane : beg;
        const uint32_t e1 = beg + 64 + lane < end ? beg + 64 + lane : beg;
        bb = g0 = g1 = g2 = g3 = g4 = g5 = make_float4(0, 0, 0, 0);
        id_next = 0;
        if (beg < end) {  // wave-uniform
            const uint32_t id0 = point_list[e0];
            id_next = point_list[e1];
            bb = bbox4[id0];
            const float4 *r = rec4 + (size_t)id0 * 6;
            g0 = r[0]; g1 = r[1]; g2 = r[2]; g3 = r[3]; g4 = r[4]; g5 = r[5];
        }
    }

    // loop-invariant lane predicates "my column / row is c" as wave masks (SGPR pairs)
    unsigned long long colsel[8], rowsel[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        colsel[c] = __builtin_amdgcn_ballot_w64((lane & 7) == c);
        rowsel[c] = __builtin_amdgcn_ballot_w64((lane >> 3) == c);
    }

    for (uint32_t base = beg; base < end; base += 64) {
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
        // ---- lanes = entries: which pixel columns / rows of the quadrant does my entry's cull box cover? ----------
        const bool valid = base + lane < end;
        unsigned long long xm[8], ym[8], xany = 0, yany = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float fx = qxlo + (float)c, fy = qylo + (float)c;
            xm[c] = __builtin_amdgcn_ballot_w64(valid && bb.x <= fx && bb.z >= fx);
            ym[c] = __builtin_amdgcn_ballot_w64(valid && bb.y <= fy && bb.w >= fy);
            xany |= xm[c];
            yany |= ym[c];
        }
        const unsigned long long hitmask = xany & yany;
        if ((hitmask >> lane) & 1ull) {
            // rebase to the quadrant origin: C' = C + (q0.x - ox)*A + (q0.y - oy)*B with o = rint(centre); centre -= q0
            const float ox = rintf(g2.y), oy = rintf(g2.z);
            const float ux = qxlo - ox, uy = qylo - oy;
            const float Cx = fmaf(uy, g0.w, fmaf(ux, g0.x, g1.z));
            const float Cy = fmaf(uy, g1.x, fmaf(ux, g0.y, g1.w));
            const float Cz = fmaf(uy, g1.y, fmaf(ux, g0.z, g2.x));
            planes[0][lane] = g0;
            planes[1][lane] = make_float4(g1.x, g1.y, Cx, Cy);
            planes[2][lane] = make_float4(Cz, g2.y - qxlo, g2.z - qylo, g2.w);
            planes[3][lane] = g3;
            planes[4][lane] = g4;
            planes[5][lane] = g5;
        }
        {   // issue the next chunk's loads (ids arrived during the previous iteration) and the ids after that
            const uint32_t idn = id_next;
            const uint32_t e2 = base + 128 + lane < end ? base + 128 + lane : beg;
            id_next = point_list[e2];
            bb = bbox4[idn];
            const float4 *r = rec4 + (size_t)idn * 6;
            g0 = r[0]; g1 = r[1]; g2 = r[2]; g3 = r[3]; g4 = r[4]; g5 = r[5];
        }
        ++stat_chunks;
        if (hitmask == 0 || (flags & 2)) continue;  // flag 2: staging only (measurement aid, not in the public header)
        // ---- lanes = pixels: my own survivor list = entries whose box covers MY column and MY row ---------------
        unsigned long long mx = xm[0], my = ym[0];
#pragma unroll
        for (int c = 1; c < 8; ++c) {
            mx = ((colsel[c] >> lane) & 1ull) ? xm[c] : mx;
            my = ((rowsel[c] >> lane) & 1ull) ? ym[c] : my;
        }
        unsigned long long m = done ? 0ull : (mx & my);
        // Lanes walk their own lists independently (compositing order only matters per pixel), kU entries per trip:
        // the kU gathers and alpha evaluations are mutually independent (one basic block, high issue rate, LDS latency
        // paid once per trip), then the contributing ones are composited in order.  An exhausted lane re-reads its
        // last slot with the pass flag forced off.
        constexpr int kU = 4;
        while (true) {
            int j[kU];
            bool live[kU];
            int last = lane;
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                live[u] = m != 0;
                j[u] = live[u] ? __builtin_ctzll(m) : last;
                last = j[u];
                m &= m - 1;
            }
            stat_iters += kU;
            Rec r[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) r[u] = lds_read_rec(planes, j[u]);
            Alpha e[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) e[u] = eval_alpha(r[u], dx, dy);
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (live[u] && e[u].pass && !done) composite(r[u], e[u], a, done);
            if (done) m = 0;
            if (__builtin_amdgcn_ballot_w64(m != 0) == 0) break;
        }
    }

    if ((flags & GA_SURFEL_FLAG_STATS) && lane == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_ITERS), (unsigned long long)stat_iters);
        atomicMax(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_MAX_ITERS), (unsigned long long)stat_iters);
        atomicAdd(reinterpret_cast<unsigned long long *>(status + GA_STATUS_BLEND_CHUNKS), (unsigned long long)stat_chunks);
    }
    if (inside) {
        const size_t HW = (size_t)dm.H * dm.W, pid = (size_t)pyi * dm.W + pxi;
        float *oc = out_color + (size_t)v * 3 * HW + pid;
        float *oo = out_others + (size_t)v * 7 * HW + pid;
        oc[0] = a.C0 + a.T * bg[0];
        oc[HW] = a.C1 + a.T * bg[1];
        oc[2 * HW] = a.C2 + a.T * bg[2];
        oo[0] = a.Dp;
        oo[HW] = 1.0f - a.T;
        oo[2 * HW] = a.N0;
        oo[3 * HW] = a.N1;
        oo[4 * HW] = a.N2;
        oo[5 * HW] = a.median;
        oo[6 * HW] = a.dist;
    }
}

void launch_blend(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s)
{
    const int nt = d.V * d.tiles;
    hipLaunchKernelGGL(surfel_blend_kernel, dim3(nt), dim3(256), 0, s, ws.tile_start, ws.tile_order, ws.point_list,
                       ws.bbox, ws.record, a.bg, d, a.out_color, a.out_others, ws.status, a.flags);
}

}  // namespace ga
