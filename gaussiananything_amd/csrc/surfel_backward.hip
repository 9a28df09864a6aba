// surfel_backward.hip -- backward pass of the 2D-surfel rasterizer, gfx950.
//
// What the reference's training call sites differentiate through: GaussianRasterizer(...)(means3D, means2D, ...) at
// /root/reference/nsr/gs_surfel.py:104-114 (upstream diff_surfel_rasterization backward.cu, third-party and absent).  The
// forward being differentiated is SURVEY.md Appendix A.1; the piecewise-constant choices are treated as constants of the
// gradient exactly as oracle/surfel_autograd.py (the backward oracle) does: the selection min(rho3d, rho2d), the
// alpha >= 1/255, depth >= near and T(1 - alpha) >= 1e-4 tests, the 0.99 clamp (no gradient above it), the sign that turns
// the normal towards the camera, and the median depth (not differentiated).
//
// Three kernels, correctness first (the forward is the hot path of this repository; this is the training-side widening):
//   1. surfel_bwd_record_kernel   per (view, Gaussian): the forward's per-splat quantities once more -- Tu, Tv, Tw, the
//      screen-space centre, the camera-facing normal, opacity, colour -- as a 24-float record;
//   2. surfel_blend_bwd_kernel    one workgroup per 16x16 tile, one thread per pixel.  Pass 1 walks the tile's depth-ordered
//      list as the forward does and leaves the totals W = sum w, M1 = sum w m, M2 = sum w m^2 and the final
//      transmittance; pass 2 walks it again with the prefix P_i = sum_{j<=i} w_j v_j and forms, per contributing pair,
//          v_i        = gC.c_i + gN.n_i + gD d_i + gDist (m_i^2 W - 2 m_i M1 + M2)      (dist = sum_{j<i} w_i w_j (m_i - m_j)^2)
//          dL/dalpha_i = T_i v_i - (V - P_i) / (1 - alpha_i) - T_final (gC.bg - gA) / (1 - alpha_i)
//          dL/ddepth_i = w_i gD + 2 gDist w_i (m_i W - M1) dm/dd
//      and from them the gradients of opacity, colour, normal, centre and of Tu / Tv / Tw through
//      p = (px Tw - Tu) x (py Tw - Tv).  A tile's list is staged 128 entries at a time; the 18 gradient words of an entry
//      are accumulated with LDS atomics and flushed with one global atomic per word and (tile, entry);
//   3. surfel_preprocess_bwd_kernel per (view, Gaussian): through M = Hm P N_pix (and the bounding-box centre formula for the
//      low-pass filter's centre), the view rotation of the normal and the normalised quaternion to means3D, scales, rotations,
//      opacities and colours, summed over the views with atomics.
#include <hip/hip_fp16.h>

#include "surfel_common.h"

namespace ga {

constexpr int kBRec = 24;    // backward record floats: Tu(3) Tv(3) Tw(3) xy(2) opa nv(3) rgb(3) | cull half-extents rx ry | pad
constexpr int kGRec = 18;    // gradient record floats: dTu(3) dTv(3) dTw(3) dxy(2) dopa dnv(3) drgb(3)
constexpr int kBwdChunk = 128;

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

__global__ __launch_bounds__(256) void surfel_bwd_record_kernel(const float *__restrict__ means3D, const float *__restrict__ opacities,
                                                                const float *__restrict__ colors, const float *__restrict__ scales,
                                                                const float *__restrict__ rotations, const float *__restrict__ viewmatrix,
                                                                const float *__restrict__ projmatrix, float scale_modifier, Dims dm,
                                                                const int32_t *__restrict__ radii, const float *__restrict__ fwd_record,
                                                                float *__restrict__ brec, float *__restrict__ grec)
{
    const int v = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
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

// one (pixel, entry) evaluation: the forward's arithmetic (oracle_blend) with what the gradient needs kept
struct PairFwd {
    float k[3], l[3], p[3], sx, sy, rho, G, raw, alpha, depth, dxc, dyc;
    bool use3d, ok;
};

__device__ __forceinline__ void pair_forward(const float *__restrict__ b, float pxf, float pyf, PairFwd &o)
{
    const float *Tu = b, *Tv = b + 3, *Tw = b + 6;
    o.ok = false;
    if (!(fabsf(pxf - b[9]) <= b[18]) || !(fabsf(pyf - b[10]) <= b[19])) return;   // outside the cull box
    for (int a = 0; a < 3; ++a) { o.k[a] = pxf * Tw[a] - Tu[a]; o.l[a] = pyf * Tw[a] - Tv[a]; }
    o.p[0] = o.k[1] * o.l[2] - o.k[2] * o.l[1];
    o.p[1] = o.k[2] * o.l[0] - o.k[0] * o.l[2];
    o.p[2] = o.k[0] * o.l[1] - o.k[1] * o.l[0];
    if (o.p[2] == 0.0f) return;
    o.sx = o.p[0] / o.p[2]; o.sy = o.p[1] / o.p[2];
    const float rho3d = o.sx * o.sx + o.sy * o.sy;
    o.dxc = b[9] - pxf; o.dyc = b[10] - pyf;
    const float rho2d = kFilterInvSquare * (o.dxc * o.dxc + o.dyc * o.dyc);
    o.use3d = rho3d <= rho2d;
    o.rho = fminf(rho3d, rho2d);
    o.depth = o.use3d ? (o.sx * Tw[0] + o.sy * Tw[1]) + Tw[2] : Tw[2];
    if (o.depth < kNear) return;
    if (-0.5f * o.rho > 0.0f) return;
    o.G = expf(-0.5f * o.rho);
    o.raw = b[11] * o.G;
    o.alpha = fminf(0.99f, o.raw);
    if (o.alpha < 1.0f / 255.0f) return;
    o.ok = true;
}

__global__ __launch_bounds__(256) void surfel_blend_bwd_kernel(const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ point_list,
                                                               const float *__restrict__ brec, const float *__restrict__ bg, Dims dm,
                                                               const float *__restrict__ out_color, const float *__restrict__ out_others,
                                                               const float *__restrict__ g_color, const float *__restrict__ g_others,
                                                               float *__restrict__ grec, const int64_t *__restrict__ status)
{
    __shared__ float srec[kBwdChunk][kBRec];
    __shared__ float sgrad[kBwdChunk][kGRec + 1];   // (+1: the 18 words of neighbouring entries start in different banks)
    __shared__ uint32_t sid[kBwdChunk];
    if (status[GA_STATUS_OVERFLOW]) return;
    const int vt = blockIdx.x, v = vt / dm.tiles, tile = vt - v * dm.tiles;
    const int tx = tile % dm.gx, ty = tile / dm.gx;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int pxi = tx * kTile + lx, pyi = ty * kTile + ly;
    const bool inside = pxi < dm.W && pyi < dm.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const uint32_t beg = tile_start[vt], end = tile_start[vt + 1];
    if (beg == end) return;
    const float kM = kFar / (kFar - kNear);
    const size_t HW = (size_t)dm.H * dm.W, pid = (size_t)pyi * dm.W + pxi;
    const size_t vbase = (size_t)v * dm.N;

    auto stage = [&](uint32_t cbeg, uint32_t cn) {
        __syncthreads();   // everybody has left the previous chunk
        for (uint32_t e = threadIdx.x; e < cn; e += 256) sid[e] = point_list[cbeg + e];
        for (uint32_t w = threadIdx.x; w < cn * (kGRec + 1); w += 256) (&sgrad[0][0])[w] = 0.0f;
        __syncthreads();
        for (uint32_t w = threadIdx.x; w < cn * kBRec; w += 256) {
            const uint32_t e = w / kBRec, f = w - e * kBRec;
            srec[e][f] = brec[(vbase + sid[e]) * kBRec + f];
        }
        __syncthreads();
    };

    // ---- pass 1: totals of the forward walk of my pixel ---------------------------------------------------------
    float T = 1.0f, W = 0.0f, M1 = 0.0f, M2 = 0.0f;
    bool done = !inside;
    for (uint32_t cbeg = beg; cbeg < end; cbeg += kBwdChunk) {
        const uint32_t cn = min((uint32_t)kBwdChunk, end - cbeg);
        if (__syncthreads_and(done)) break;
        stage(cbeg, cn);
        for (uint32_t e = 0; e < cn && !done; ++e) {
            PairFwd f;
            pair_forward(srec[e], pxf, pyf, f);
            if (!f.ok) continue;
            const float test_T = T * (1.0f - f.alpha);
            if (test_T < 0.0001f) { done = true; break; }
            const float w = f.alpha * T, m = kM * (1.0f - kNear / f.depth);
            W += w; M1 += m * w; M2 += m * m * w;
            T = test_T;
        }
    }
    const float T_final = T;
    float gC[3] = {0.f, 0.f, 0.f}, gN[3] = {0.f, 0.f, 0.f}, gD = 0.f, gA = 0.f, gDist = 0.f, Vtot = 0.f;
    if (inside) {
        for (int c = 0; c < 3; ++c) gC[c] = g_color[((size_t)v * 3 + c) * HW + pid];
        const float *go = g_others + (size_t)v * 7 * HW + pid, *oo = out_others + (size_t)v * 7 * HW + pid;
        gD = go[0]; gA = go[HW]; gN[0] = go[2 * HW]; gN[1] = go[3 * HW]; gN[2] = go[4 * HW]; gDist = go[6 * HW];
        const float *oc = out_color + (size_t)v * 3 * HW + pid;
        // V = sum w v = gC.(C - T bg) + gN.N + gD Dp + 2 gDist (W M2 - M1^2)
        for (int c = 0; c < 3; ++c) Vtot += gC[c] * (oc[c * HW] - T_final * bg[c]);
        Vtot += gN[0] * oo[2 * HW] + gN[1] * oo[3 * HW] + gN[2] * oo[4 * HW] + gD * oo[0] + 2.0f * gDist * (W * M2 - M1 * M1);
    }
    const float bgterm = (gC[0] * bg[0] + gC[1] * bg[1] + gC[2] * bg[2]) - gA;

    // ---- pass 2: gradients ---------------------------------------------------------------------------------------
    T = 1.0f;
    float P = 0.0f;
    done = !inside;
    for (uint32_t cbeg = beg; cbeg < end; cbeg += kBwdChunk) {
        const uint32_t cn = min((uint32_t)kBwdChunk, end - cbeg);
        if (__syncthreads_and(done)) break;
        stage(cbeg, cn);
        for (uint32_t e = 0; e < cn && !done; ++e) {
            const float *b = srec[e];
            PairFwd f;
            pair_forward(b, pxf, pyf, f);
            if (!f.ok) continue;
            const float test_T = T * (1.0f - f.alpha);
            if (test_T < 0.0001f) { done = true; break; }
            const float w = f.alpha * T, m = kM * (1.0f - kNear / f.depth);
            const float Dq = m * m * W - 2.0f * m * M1 + M2;
            const float vi = (gC[0] * b[15] + gC[1] * b[16] + gC[2] * b[17]) + (gN[0] * b[12] + gN[1] * b[13] + gN[2] * b[14]) +
                             gD * f.depth + gDist * Dq;
            P += w * vi;
            const float inv1ma = 1.0f / (1.0f - f.alpha);
            const float dL_dalpha = T * vi - (Vtot - P) * inv1ma - T_final * bgterm * inv1ma;
            const float dL_ddepth = w * gD + 2.0f * gDist * w * (m * W - M1) * (kM * kNear / (f.depth * f.depth));
            float *g = sgrad[e];
            for (int c = 0; c < 3; ++c) { atomicAdd(g + 15 + c, w * gC[c]); atomicAdd(g + 12 + c, w * gN[c]); }
            const float dL_draw = f.raw > 0.99f ? 0.0f : dL_dalpha;       // the clamp carries no gradient
            atomicAdd(g + 11, f.G * dL_draw);
            const float dL_drho = -0.5f * f.raw * dL_draw;
            float dsx = 0.f, dsy = 0.f, dTw[3] = {0.f, 0.f, dL_ddepth};
            if (f.use3d) {
                dsx = 2.0f * f.sx * dL_drho + dL_ddepth * b[6];
                dsy = 2.0f * f.sy * dL_drho + dL_ddepth * b[7];
                dTw[0] = dL_ddepth * f.sx; dTw[1] = dL_ddepth * f.sy;
            } else {
                atomicAdd(g + 9, 2.0f * kFilterInvSquare * f.dxc * dL_drho);
                atomicAdd(g + 10, 2.0f * kFilterInvSquare * f.dyc * dL_drho);
            }
            // s = p.xy / p.z ; p = k x l ; k = px Tw - Tu ; l = py Tw - Tv
            const float ipz = 1.0f / f.p[2];
            const float gp[3] = {dsx * ipz, dsy * ipz, -(dsx * f.sx + dsy * f.sy) * ipz};
            const float dk[3] = {f.l[1] * gp[2] - f.l[2] * gp[1], f.l[2] * gp[0] - f.l[0] * gp[2], f.l[0] * gp[1] - f.l[1] * gp[0]};
            const float dl[3] = {gp[1] * f.k[2] - gp[2] * f.k[1], gp[2] * f.k[0] - gp[0] * f.k[2], gp[0] * f.k[1] - gp[1] * f.k[0]};
            for (int a = 0; a < 3; ++a) {
                atomicAdd(g + a, -dk[a]);
                atomicAdd(g + 3 + a, -dl[a]);
                atomicAdd(g + 6 + a, dTw[a] + pxf * dk[a] + pyf * dl[a]);
            }
            T = test_T;
        }
        __syncthreads();   // the chunk's gradient words are complete: one global atomic per word
        for (uint32_t w2 = threadIdx.x; w2 < cn * kGRec; w2 += 256) {
            const uint32_t e = w2 / kGRec, f2 = w2 - e * kGRec;
            const float val = sgrad[e][f2];
            if (val != 0.0f) atomicAdd(grec + (vbase + sid[e]) * kGRec + f2, val);
        }
    }
}

__global__ __launch_bounds__(256) void surfel_preprocess_bwd_kernel(const float *__restrict__ means3D, const float *__restrict__ scales,
                                                                    const float *__restrict__ rotations, const float *__restrict__ viewmatrix,
                                                                    const float *__restrict__ projmatrix, float scale_modifier, Dims dm,
                                                                    const int32_t *__restrict__ radii, const float *__restrict__ grec,
                                                                    float *__restrict__ d_means, float *__restrict__ d_opac,
                                                                    float *__restrict__ d_colors, float *__restrict__ d_scales,
                                                                    float *__restrict__ d_rot)
{
    const int v = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dm.N) return;
    const size_t idx = (size_t)v * dm.N + i;
    if (radii[idx] <= 0) return;
    const float *g = grec + idx * kGRec;
    const float *vm = viewmatrix + 16 * v, *pm = projmatrix + 16 * v;
    SplatFwd s;
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
    const float halfW = (float)dm.W / 2.0f, halfH = (float)dm.H / 2.0f;
    const float cW = (float)(dm.W - 1) / 2.0f, cH = (float)(dm.H - 1) / 2.0f;
    float dHm[3][3];
    for (int a = 0; a < 3; ++a) {
        const float dA[4] = {dM[a][0] * halfW, dM[a][1] * halfH, 0.0f, dM[a][0] * cW + dM[a][1] * cH + dM[a][2]};
        for (int j = 0; j < 3; ++j) dHm[a][j] = dA[0] * pm[4 * j] + dA[1] * pm[4 * j + 1] + dA[2] * pm[4 * j + 2] + dA[3] * pm[4 * j + 3];
    }
    float dR[3][3];   // dL/dR[row][col]: columns tu, tv, n
    float dsu = 0.f, dsv = 0.f;
    for (int r = 0; r < 3; ++r) {
        dR[r][0] = s.su * dHm[0][r];
        dR[r][1] = s.sv * dHm[1][r];
        dsu += s.tu[r] * dHm[0][r];
        dsv += s.tv[r] * dHm[1][r];
        dR[r][2] = s.mult * (g[12] * vm[4 * r] + g[13] * vm[4 * r + 1] + g[14] * vm[4 * r + 2]);   // nv_c = mult sum_r vm[4 r + c] n_r
    }
    for (int a = 0; a < 3; ++a) atomicAdd(d_means + 3 * i + a, dHm[2][a]);
    atomicAdd(d_scales + 2 * i, scale_modifier * dsu);
    atomicAdd(d_scales + 2 * i + 1, scale_modifier * dsv);
    atomicAdd(d_opac + i, g[11]);
    for (int a = 0; a < 3; ++a) atomicAdd(d_colors + 3 * i + a, g[15 + a]);
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
    for (int c = 0; c < 4; ++c) atomicAdd(d_rot + 4 * i + c, (dq[c] - qh[c] * dotq) * s.qs);
}

}  // namespace ga

extern "C" int ga_surfel_backward(const GaSurfelBackwardArgs *a, void *stream_v)
{
    using namespace ga;
    if (!a) return GA_ERR_NULL_ARG;
    const GaSurfelForwardArgs &f = a->fwd;
    Dims d;
    const int64_t gx = ((int64_t)f.image_width + kTile - 1) / kTile, gy = ((int64_t)f.image_height + kTile - 1) / kTile;
    if (f.num_points < 0 || f.num_views <= 0 || f.image_height <= 0 || f.image_width <= 0) return GA_ERR_BAD_SHAPE;
    d.N = f.num_points; d.V = f.num_views; d.H = f.image_height; d.W = f.image_width; d.gx = (int)gx; d.gy = (int)gy; d.tiles = (int)(gx * gy);
    GaSurfelWorkspaceLayout L;
    const int rc = ga_surfel_workspace_layout(d.N, d.V, d.H, d.W, f.capacity, &L);
    if (rc != GA_OK) return rc;
    if (!f.workspace || f.workspace_bytes < L.total_bytes || !f.out_color || !f.out_others || !f.radii || !f.bg || !f.viewmatrix ||
        !f.projmatrix || !a->grad_color || !a->grad_others || !a->scratch || !a->grad_means3D || !a->grad_opacities ||
        !a->grad_colors || !a->grad_scales || !a->grad_rotations)
        return GA_ERR_NULL_ARG;
    if (d.N > 0 && (!f.means3D || !f.opacities || !f.colors || !f.scales || !f.rotations)) return GA_ERR_NULL_ARG;
    if (a->scratch_bytes < ga_surfel_backward_scratch_bytes(d.N, d.V)) return GA_ERR_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_v);
    (void)hipGetLastError();
    if (d.N == 0) return GA_OK;
    unsigned char *w = static_cast<unsigned char *>(f.workspace);
    const int64_t *status = reinterpret_cast<const int64_t *>(w + L.status);
    const uint32_t *tile_start = reinterpret_cast<const uint32_t *>(w + L.tile_start);
    const uint32_t *point_list = reinterpret_cast<const uint32_t *>(w + L.point_list);
    float *brec = static_cast<float *>(a->scratch);
    float *grec = brec + (size_t)d.N * d.V * kBRec;
    (void)hipMemsetAsync(a->grad_means3D, 0, (size_t)d.N * 3 * 4, s);
    (void)hipMemsetAsync(a->grad_opacities, 0, (size_t)d.N * 4, s);
    (void)hipMemsetAsync(a->grad_colors, 0, (size_t)d.N * 3 * 4, s);
    (void)hipMemsetAsync(a->grad_scales, 0, (size_t)d.N * 2 * 4, s);
    (void)hipMemsetAsync(a->grad_rotations, 0, (size_t)d.N * 4 * 4, s);
    const dim3 gridN((unsigned)((d.N + 255) / 256), (unsigned)d.V);
    hipLaunchKernelGGL(surfel_bwd_record_kernel, gridN, dim3(256), 0, s, f.means3D, f.opacities, f.colors, f.scales, f.rotations,
                       f.viewmatrix, f.projmatrix, f.scale_modifier, d, f.radii, reinterpret_cast<const float *>(w + L.record), brec,
                       grec);
    hipLaunchKernelGGL(surfel_blend_bwd_kernel, dim3((unsigned)(d.V * d.tiles)), dim3(256), 0, s, tile_start, point_list, brec, f.bg, d,
                       f.out_color, f.out_others, a->grad_color, a->grad_others, grec, status);
    hipLaunchKernelGGL(surfel_preprocess_bwd_kernel, gridN, dim3(256), 0, s, f.means3D, f.scales, f.rotations, f.viewmatrix,
                       f.projmatrix, f.scale_modifier, d, f.radii, grec, a->grad_means3D, a->grad_opacities, a->grad_colors,
                       a->grad_scales, a->grad_rotations);
    return hipGetLastError() == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

extern "C" size_t ga_surfel_backward_scratch_bytes(int32_t num_points, int32_t num_views)
{
    return (size_t)num_points * (size_t)num_views * (ga::kBRec + ga::kGRec) * sizeof(float);
}
