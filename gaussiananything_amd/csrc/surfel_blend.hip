// surfel_blend.hip -- per-tile front-to-back alpha compositing of colour + depth + normal (+ median depth,
// distortion), gfx950.  Replaces upstream renderCUDA of diff_surfel_rasterization (call site
// /root/reference/nsr/gs_surfel.py:100-114; consumer of the 7 allmap channels :121-142); arithmetic per
// SURVEY.md Appendix A.1 "Blend".
//
// MI355X-first formulation ("wave-autonomous" blend), not the CUDA block-cooperative one:
//   * a 256-thread workgroup still owns one 16x16 tile (that granularity is part of the semantics: the tile rect
//     decides which pixels a splat may touch), but each of its four 64-lane wavefronts owns one 8x8 quadrant and runs
//     completely on its own -- no LDS staging, no workgroup barrier, independent early termination;
//   * the tile's depth-ordered list is consumed 64 entries at a time with lanes = entries: every lane fetches one
//     entry's conservative {alpha >= 1/255} pixel box (16 B) and tests it against the quadrant; a 64-bit ballot is
//     the list of entries that can contribute.  For the small splats of real scenes this removes about half of
//     the (pixel, splat) evaluations, and it shortens the serial per-tile chain, which -- not HBM -- is the critical
//     path;
//   * survivors are walked with s_ff1 on the ballot; the entry index is then wave-uniform, so its 80-byte record is
//     fetched through the SCALAR path (s_load_dwordx4 into SGPRs): per-splat data never occupies VGPRs or LDS
//     bandwidth, and the inner loop is pure VALU with lanes = pixels;
//   * rejected lanes are handled by EXEC masking; a wave leaves the list as soon as its 64 pixels are saturated.
// Pixel results are compared with the oracle by MSE (<= 1e-5, tests/), so this TU may contract to FMA and uses
// v_rcp_f32 / v_exp_f32 instead of IEEE division and libm expf.
#include "surfel_common.h"

namespace ga {

struct PixelAcc {
    float T, C0, C1, C2, N0, N1, N2, Dp, M1, M2, dist, median;
};

__global__ __launch_bounds__(256) void surfel_blend_kernel(const uint32_t *__restrict__ tile_start,
                                                           const uint32_t *__restrict__ point_list,
                                                           const float *__restrict__ bbox,
                                                           const float *__restrict__ record,
                                                           const float *__restrict__ bg, Dims dm,
                                                           float *__restrict__ out_color,
                                                           float *__restrict__ out_others,
                                                           const int64_t *__restrict__ status)
{
    if (status[GA_STATUS_OVERFLOW]) return;
    const uint32_t nt = (uint32_t)(dm.V * dm.tiles);
    const uint32_t vt = xcd_remap(blockIdx.x, nt);
    const int v = (int)(vt / (uint32_t)dm.tiles), tile = (int)(vt - (uint32_t)v * dm.tiles);
    const int tx = tile % dm.gx, ty = tile / dm.gx;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qx0 = tx * kTile + (wave & 1) * 8, qy0 = ty * kTile + (wave >> 1) * 8;
    if (qx0 >= dm.W || qy0 >= dm.H) return;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < dm.W && pyi < dm.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const float qxlo = (float)qx0, qxhi = (float)(qx0 + 7), qylo = (float)qy0, qyhi = (float)(qy0 + 7);

    const uint32_t beg = tile_start[vt], end = tile_start[vt + 1];
    const size_t vbase = (size_t)v * dm.N;
    const float4 *__restrict__ bbox4 = reinterpret_cast<const float4 *>(bbox) + vbase;
    const float *__restrict__ recv = record + vbase * kRec;

    PixelAcc a = {1.0f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool done = !inside;
    const float kM = kFar / (kFar - kNear);

    for (uint32_t base = beg; base < end; base += 64) {
        if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
        const uint32_t e = base + lane;
        uint32_t id = 0;
        bool hit = false;
        if (e < end) {
            id = point_list[e];
            const float4 bb = bbox4[id];
            hit = bb.x <= qxhi && bb.z >= qxlo && bb.y <= qyhi && bb.w >= qylo;
        }
        unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const uint32_t sid = (uint32_t)__builtin_amdgcn_readlane((int)id, j);
            const float4 *__restrict__ r = reinterpret_cast<const float4 *>(recv + (size_t)sid * kRec);
            const float4 r0 = r[0], r1 = r[1], r2 = r[2];
            const float Tux = r0.x, Tuy = r0.y, Tuz = r0.z, Tvx = r0.w, Tvy = r1.x, Tvz = r1.y;
            const float Twx = r1.z, Twy = r1.w, Twz = r2.x, cx = r2.y, cy = r2.z, opa = r2.w;
            if (!done) {
                const float kx = pxf * Twx - Tux, ky = pxf * Twy - Tuy, kz = pxf * Twz - Tuz;
                const float lx = pyf * Twx - Tvx, ly = pyf * Twy - Tvy, lz = pyf * Twz - Tvz;
                const float p0 = ky * lz - kz * ly, p1 = kz * lx - kx * lz, p2 = kx * ly - ky * lx;
                if (p2 != 0.0f) {
                    const float rz = __builtin_amdgcn_rcpf(p2);
                    const float sx = p0 * rz, sy = p1 * rz;
                    const float rho3d = sx * sx + sy * sy;
                    const float dx = cx - pxf, dy = cy - pyf;
                    const float rho2d = kFilterInvSquare * (dx * dx + dy * dy);
                    const float rho = fminf(rho3d, rho2d);
                    const float depth = (rho3d <= rho2d) ? (sx * Twx + sy * Twy) + Twz : Twz;
                    // power = -0.5*rho ; (power > 0) <=> rho < 0 ; NaN rho fails both tests below like upstream's min()
                    if (!(depth < kNear) && !(rho < 0.0f)) {
                        const float alpha = fminf(0.99f, opa * __builtin_amdgcn_exp2f(rho * -0.72134752044f));
                        if (alpha >= 1.0f / 255.0f) {
                            const float test_T = a.T * (1.0f - alpha);
                            if (test_T < 0.0001f) {
                                done = true;
                            } else {
                                const float4 r3 = r[3], r4 = r[4];
                                const float w = alpha * a.T;
                                const float A = 1.0f - a.T;
                                const float m = kM * (1.0f - kNear * __builtin_amdgcn_rcpf(depth));
                                a.dist += (m * m * A + a.M2 - 2.0f * m * a.M1) * w;
                                a.Dp += depth * w;
                                a.M1 += m * w;
                                a.M2 += m * m * w;
                                if (a.T > 0.5f) a.median = depth;
                                a.N0 += r3.x * w; a.N1 += r3.y * w; a.N2 += r3.z * w;
                                a.C0 += r4.x * w; a.C1 += r4.y * w; a.C2 += r4.z * w;
                                a.T = test_T;
                            }
                        }
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(!done) == 0) { mask = 0; }
        }
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
    hipLaunchKernelGGL(surfel_blend_kernel, dim3(nt), dim3(256), 0, s, ws.tile_start, ws.point_list, ws.bbox,
                       ws.record, a.bg, d, a.out_color, a.out_others, ws.status);
}

}  // namespace ga
