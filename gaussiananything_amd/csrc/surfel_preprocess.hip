// surfel_preprocess.hip -- per-(view, Gaussian) preprocess of the 2D-surfel rasterizer, gfx950.
//
// Replaces upstream preprocessCUDA (+ compute_transmat / compute_aabb / getRect) of diff_surfel_rasterization, the
// extension called at /root/reference/nsr/gs_surfel.py:100-114; algorithm per SURVEY.md Appendix A.1 steps 1-7.
//
// This translation unit is built with -ffp-contract=off: radii, tile rects and depth keys must be BIT-IDENTICAL to
// the oracle (oracle/surfel_raster.c), so every expression below keeps the oracle's operation order, one IEEE
// rounding per operation, correctly rounded division and sqrt (hipcc default), no FMA.
//
// MI355X notes: one thread per Gaussian x a group of views (round 5; one thread per (view, Gaussian) before), a wave reads 64
// consecutive Gaussians (768 contiguous bytes of means3D) once and keeps the camera-independent part in registers.  Outputs are written as ONE 96-byte record per splat
// (what the blend kernel stages into LDS) plus the small SoA side arrays the binning passes stream (depth, rect,
// bbox).  Tile occupancy is counted here (LDS histogram per workgroup, then L2 atomics), which replaces upstream's
// tiles_touched array + device-wide inclusive scan.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <algorithm>

#include "surfel_common.h"

#pragma clang fp contract(off)

namespace ga {

__device__ __forceinline__ int f2i(float f) { return (int)f; }  // v_cvt_i32_f32: toward zero, saturating, NaN -> 0

// What a thread keeps of ITS Gaussian for all the views it serves (round 5): everything of A.1 steps 2-3 and of the cull box that
// does not depend on the camera.  Round 4 had one thread per (view, Gaussian): the quaternion normalisation (an IEEE division and a
// square root), the rotation, the scaled tangents, the logarithm of the cull box and five input round trips were redone for each of
// the V views.  The operations and their order are unchanged, so radii / rects / depth keys stay bit-identical.
struct SplatConst {
    float px, py, pz;
    float hu[3], hv[3];      // su * tu, sv * tv (rows 0 and 1 of the homography's 3 x 4 factor Hm)
    float nn[3];             // normal R[:, 2]
    float opa, c2, r2;       // opacity; cull box: c2 = 2 ln(255 opa) with its margins, radius of the low-pass disc
    float col[3];
};

__device__ __forceinline__ SplatConst splat_constants(const float *__restrict__ means3D, const float *__restrict__ opacities,
                                                      const float *__restrict__ colors, const float *__restrict__ scales,
                                                      const float *__restrict__ rotations, float scale_modifier, int i)
{
    SplatConst c;
    c.px = means3D[3 * i]; c.py = means3D[3 * i + 1]; c.pz = means3D[3 * i + 2];
    // A.1 step 2: the quaternion is re-normalised first (upstream quat_to_rotmat); same operation order as the oracle
    const float4 q = *reinterpret_cast<const float4 *>(rotations + 4 * i);
    const float2 sc = *reinterpret_cast<const float2 *>(scales + 2 * i);
    c.opa = opacities[i];
    c.col[0] = colors[3 * i]; c.col[1] = colors[3 * i + 1]; c.col[2] = colors[3 * i + 2];
    const float qs = 1.0f / sqrtf(((q.w * q.w + q.x * q.x) + q.y * q.y) + q.z * q.z);
    const float r = q.x * qs, x = q.y * qs, y = q.z * qs, z = q.w * qs;
    const float tu[3] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y + r * z), 2.f * (x * z - r * y)};
    const float tv[3] = {2.f * (x * y - r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + r * x)};
    c.nn[0] = 2.f * (x * z + r * y); c.nn[1] = 2.f * (y * z - r * x); c.nn[2] = 1.f - 2.f * (x * x + y * y);
    const float su = scale_modifier * sc.x, sv = scale_modifier * sc.y;
#pragma unroll
    for (int k = 0; k < 3; ++k) { c.hu[k] = tu[k] * su; c.hv[k] = tv[k] * sv; }
    // cull box (see preprocess_view): not part of the bit-exact artefacts -- conservative, with margins that cover v_log / v_sqrt
    c.c2 = (2.0f * __logf(255.0f * c.opa)) * 1.002f + 0.004f;
    c.r2 = __builtin_amdgcn_sqrtf(0.5f * c.c2) * 1.001f + 0.01f;
    return c;
}

// One (view, Gaussian).  `tc` is the tile-counter array of this view: the workgroup's LDS histogram (kLds) or the global counters.
// Returns true when the record was produced; it is left in `stg` (this lane's 96-byte LDS slot) for the wave's coalesced copy-out.
__device__ __forceinline__ bool preprocess_view(const SplatConst &c, const float *__restrict__ vm, const float *__restrict__ pm,
                                                const Dims &dm, int64_t idx, int32_t *__restrict__ radii,
                                                uint16_t *__restrict__ rect_out, float *__restrict__ depth_out, float4 *stg,
                                                uint32_t *tc)
{
    const float px = c.px, py = c.py, pz = c.pz;
    int radius_i = 0;
    ushort4 rc = make_ushort4(0, 0, 0, 0);
    bool live = false;
    do {   // (one exit: the zero radius / rect of a culled splat are stored once, below)
        const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        if (vz <= 0.2f) break;

        const float halfW = (float)dm.W / 2.0f, halfH = (float)dm.H / 2.0f;
        const float cW = (float)(dm.W - 1) / 2.0f, cH = (float)(dm.H - 1) / 2.0f;
        const float Hm[3][3] = {{c.hu[0], c.hu[1], c.hu[2]}, {c.hv[0], c.hv[1], c.hv[2]}, {px, py, pz}};
        float M[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float A[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = Hm[a][0] * pm[0 + j] + Hm[a][1] * pm[4 + j] + Hm[a][2] * pm[8 + j];
                if (a == 2) s = s + pm[12 + j];
                A[j] = s;
            }
            M[a][0] = A[0] * halfW + A[3] * cW;
            M[a][1] = A[1] * halfH + A[3] * cH;
            M[a][2] = A[3];
        }
        const float Tu[3] = {M[0][0], M[1][0], M[2][0]};
        const float Tv[3] = {M[0][1], M[1][1], M[2][1]};
        const float Tw[3] = {M[0][2], M[1][2], M[2][2]};

        float nvx = vm[0] * c.nn[0] + vm[4] * c.nn[1] + vm[8] * c.nn[2];
        float nvy = vm[1] * c.nn[0] + vm[5] * c.nn[1] + vm[9] * c.nn[2];
        float nvz = vm[2] * c.nn[0] + vm[6] * c.nn[1] + vm[10] * c.nn[2];
        const float cs = -((vx * nvx + vy * nvy) + vz * nvz);
        if (cs == 0.0f) break;
        const float mult = cs > 0.0f ? 1.0f : -1.0f;
        nvx = mult * nvx; nvy = mult * nvy; nvz = mult * nvz;

        const float t0 = kCutoff * kCutoff, t1 = kCutoff * kCutoff, t2 = -1.0f;
        const float d = (t0 * (Tw[0] * Tw[0]) + t1 * (Tw[1] * Tw[1])) + t2 * (Tw[2] * Tw[2]);
        if (d == 0.0f) break;
        const float inv = 1.0f / d;
        const float f0 = inv * t0, f1 = inv * t1, f2 = inv * t2;
        const float cx = (f0 * (Tu[0] * Tw[0]) + f1 * (Tu[1] * Tw[1])) + f2 * (Tu[2] * Tw[2]);
        const float cy = (f0 * (Tv[0] * Tw[0]) + f1 * (Tv[1] * Tw[1])) + f2 * (Tv[2] * Tw[2]);
        const float hx0 = cx * cx - ((f0 * (Tu[0] * Tu[0]) + f1 * (Tu[1] * Tu[1])) + f2 * (Tu[2] * Tu[2]));
        const float hy0 = cy * cy - ((f0 * (Tv[0] * Tv[0]) + f1 * (Tv[1] * Tv[1])) + f2 * (Tv[2] * Tv[2]));
        const float ex = sqrtf(fmaxf(1e-4f, hx0)), ey = sqrtf(fmaxf(1e-4f, hy0));
        const float radius = ceilf(fmaxf(fmaxf(ex, ey), kCutoff * kFilterSize));

        const int rminx = min(dm.gx, max(0, f2i((cx - radius) / kTile)));
        const int rminy = min(dm.gy, max(0, f2i((cy - radius) / kTile)));
        const int rmaxx = min(dm.gx, max(0, f2i((cx + radius + kTile - 1) / kTile)));
        const int rmaxy = min(dm.gy, max(0, f2i((cy + radius + kTile - 1) / kTile)));
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) break;

        live = true;
        radius_i = f2i(radius);
        depth_out[idx] = vz;
        rc = make_ushort4((unsigned short)rminx, (unsigned short)rminy, (unsigned short)rmaxx, (unsigned short)rmaxy);

        const float opa = c.opa;
        // plane-form coefficients of the ray/splat intersection (see surfel_common.h), taken about the integer pixel
        // nearest the splat centre so that no term carries the ~W/2 screen offset (same accuracy class as upstream's
        // k = px*Tw - Tu, whose subtraction removes that offset per pixel):  p = (px-ox)*A + (py-oy)*B + C
        const float ox = rintf(cx), oy = rintf(cy);
        const float Uc[3] = {Tu[0] - ox * Tw[0], Tu[1] - ox * Tw[1], Tu[2] - ox * Tw[2]};
        const float Vc[3] = {Tv[0] - oy * Tw[0], Tv[1] - oy * Tw[1], Tv[2] - oy * Tw[2]};
        const float Ax = Vc[1] * Tw[2] - Vc[2] * Tw[1], Ay = Vc[2] * Tw[0] - Vc[0] * Tw[2], Az = Vc[0] * Tw[1] - Vc[1] * Tw[0];
        const float Bx = Tw[1] * Uc[2] - Tw[2] * Uc[1], By = Tw[2] * Uc[0] - Tw[0] * Uc[2], Bz = Tw[0] * Uc[1] - Tw[1] * Uc[0];
        const float Cx = Uc[1] * Vc[2] - Uc[2] * Vc[1], Cy = Uc[2] * Vc[0] - Uc[0] * Vc[2], Cz = Uc[0] * Vc[1] - Uc[1] * Vc[0];
        // Conservative pixel box of {alpha >= 1/255}: the blend loop rejects (pixel, splat) pairs outside it without
        // evaluating them.  alpha = min(.99, opa*exp(-rho/2)) >= 1/255  <=>  rho = min(rho3d, rho2d) <= c2 with
        // c2 = 2 ln(255 opa): union of the low-pass disc (rho2d <= c2: radius sqrt(c2/2) about (cx, cy)) and the projected
        // c-sigma ellipse (rho3d <= c2: same AABB construction as above with cutoff^2 = c2).  The box decides how many pairs
        // the blend evaluates (a 0.5 px margin on a 2.35 px radius cost 34 % more evaluations, tools/blend_sim.py), so the
        // margins are only what the arithmetic needs: the half-extent of the ellipse is taken about ITS centre
        // (U = Tu - bx*Tw: no difference of two ~W^2 terms as in the centre^2 - sum form), leaving relative errors ~1e-5.
        // Anything doubtful falls back to "everything".  Stored as half-extents about (cx, cy), rounded up to fp16.
        // Round 5: v_rcp_f32 / v_sqrt_f32 (1 ulp) here instead of the correctly rounded division and square roots -- the box is not
        // one of the bit-exact artefacts and its margins (2e-3 relative + 0.02 px) are four orders of magnitude above an ulp.
        const float kInf = __builtin_inff();
        float rx = kInf, ry = kInf;
        if (opa < 1.0f / 255.0f) {
            rx = ry = -1.0f;                                 // can never pass the alpha threshold
        } else {
            const float c2 = c.c2;
            const float dd = (c2 * (Tw[0] * Tw[0]) + c2 * (Tw[1] * Tw[1])) - (Tw[2] * Tw[2]);
            if (c2 < 1e30f && dd < 0.0f) {
                const float iv = __builtin_amdgcn_rcpf(dd);
                const float g0 = iv * c2, g2 = -iv;
                const float bx = (g0 * (Tu[0] * Tw[0]) + g0 * (Tu[1] * Tw[1])) + g2 * (Tu[2] * Tw[2]);
                const float by = (g0 * (Tv[0] * Tw[0]) + g0 * (Tv[1] * Tw[1])) + g2 * (Tv[2] * Tw[2]);
                const float Ux[3] = {Tu[0] - bx * Tw[0], Tu[1] - bx * Tw[1], Tu[2] - bx * Tw[2]};
                const float Uy[3] = {Tv[0] - by * Tw[0], Tv[1] - by * Tw[1], Tv[2] - by * Tw[2]};
                const float hx = -((g0 * (Ux[0] * Ux[0]) + g0 * (Ux[1] * Ux[1])) + g2 * (Ux[2] * Ux[2]));
                const float hy = -((g0 * (Uy[0] * Uy[0]) + g0 * (Uy[1] * Uy[1])) + g2 * (Uy[2] * Uy[2]));
                const float e3x = __builtin_amdgcn_sqrtf(fmaxf(hx, 0.0f)) * 1.002f + 0.02f;
                const float e3y = __builtin_amdgcn_sqrtf(fmaxf(hy, 0.0f)) * 1.002f + 0.02f;
                const float r2 = c.r2;
                const float xmin = fminf(bx - e3x, cx - r2), xmax = fmaxf(bx + e3x, cx + r2);
                const float ymin = fminf(by - e3y, cy - r2), ymax = fmaxf(by + e3y, cy + r2);
                if (xmin == xmin && xmax == xmax && ymin == ymin && ymax == ymax && hx == hx && hy == hy) {
                    rx = fmaxf(cx - xmin, xmax - cx);
                    ry = fmaxf(cy - ymin, ymax - cy);
                }
            }
        }
        // fp16, rounded towards +inf (a value beyond the fp16 range becomes +inf = no bound)
        const uint32_t cull = (uint32_t)__half_as_ushort(__float2half_ru(rx)) |
                              ((uint32_t)__half_as_ushort(__float2half_ru(ry)) << 16);
        float4 *rec = stg;
        rec[0] = make_float4(Ax, Ay, Bx, By);
        rec[1] = make_float4(Cx, Cy, Az, Bz);
        rec[2] = make_float4(cx, cy, Cz, opa);
        rec[3] = make_float4(Tw[0], Tw[1], Tw[2], __uint_as_float(cull));
        rec[4] = make_float4(nvx, nvy, nvz, c.col[0]);
        rec[5] = make_float4(c.col[1], c.col[2], 0.0f, 0.0f);

        for (int ty = rminy; ty < rmaxy; ++ty)
            for (int tx = rminx; tx < rmaxx; ++tx) atomicAdd(tc + ty * dm.gx + tx, 1u);
    } while (false);
    radii[idx] = radius_i;
    *reinterpret_cast<ushort4 *>(rect_out + 4 * idx) = rc;
    return live;
}

// One workgroup = 256 threads = 256 consecutive Gaussians x `vg` consecutive views (blockIdx.y * vg ...): a thread computes its
// Gaussian's camera-independent part once and walks the views.  Tile occupancy is accumulated in LDS histograms of the views'
// tiles (vg * tiles words, <= 32 KiB) and flushed with one global atomic per touched tile per workgroup: the hottest tile of a real
// scene receives thousands of increments per view and same-address L2 atomics serialise (measured 0.27 ms for 1.4 M increments,
// profiles/r1a_*).  Views with more than kLdsTiles tiles use the global counters directly.
template <bool kLds, int SPT>
__global__ __launch_bounds__(256) void surfel_preprocess_kernel(
    const float *__restrict__ means3D, const float *__restrict__ opacities, const float *__restrict__ colors,
    const float *__restrict__ scales, const float *__restrict__ rotations, const float *__restrict__ viewmatrix,
    const float *__restrict__ projmatrix, float scale_modifier, Dims dm, int vg, int nt, int32_t *__restrict__ radii,
    uint16_t *__restrict__ rect_out, float *__restrict__ depth_out,
    float *__restrict__ rec_out, uint32_t *__restrict__ tile_count, unsigned long long *__restrict__ view_total)
{
    extern __shared__ uint32_t hist[];
    // Records leave through LDS: a lane's record is 96 contiguous bytes, so direct stores would be six 16-byte pieces
    // at a 96-byte stride per instruction (partial lines); the wave's 64 records are one contiguous 6 KiB block, written
    // with six fully coalesced 1 KiB stores instead.
    __shared__ __attribute__((aligned(16))) float4 stage[256 * (kRec / 4)];
    const int v0 = (int)blockIdx.y * vg, nv = min(vg, dm.V - v0);
    if (kLds) {
        for (int t = threadIdx.x; t < nv * dm.tiles; t += 256) hist[t] = 0;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *wstage = stage + wave * 64 * (kRec / 4);
    for (int sp = 0; sp < SPT; ++sp) {   // SPT consecutive groups of 256 Gaussians: the LDS histograms aggregate over all of them
        const int i = ((int)blockIdx.x * SPT + sp) * 256 + (int)threadIdx.x;
        const bool mine = i < dm.N;
        if (__builtin_amdgcn_ballot_w64(mine) == 0) break;
        SplatConst c;
        if (mine) c = splat_constants(means3D, opacities, colors, scales, rotations, scale_modifier, i);
        for (int k = 0; k < nv; ++k) {
            const int v = v0 + k;
            const float *vm = viewmatrix + 16 * v, *pm = projmatrix + 16 * v;
            bool live = false;
            if (mine)
                live = preprocess_view(c, vm, pm, dm, (int64_t)v * dm.N + i, radii, rect_out, depth_out, wstage + lane * (kRec / 4),
                                       kLds ? hist + (size_t)k * dm.tiles : tile_count + (size_t)v * dm.tiles);
            if (__builtin_amdgcn_ballot_w64(live) != 0) {  // wave-uniform; records of culled lanes are never read
                const int first = i - lane;                                    // first Gaussian of this wave
                const int nq = min(64, dm.N - first) * (kRec / 4);             // float4s inside the array
                float4 *dst = reinterpret_cast<float4 *>(rec_out + ((size_t)v * dm.N + first) * kRec);
#pragma unroll
                for (int j = 0; j < kRec / 4; ++j) {
                    const int q = lane + 64 * j;
                    if (q < nq) {
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        if (nt) __builtin_nontemporal_store(*reinterpret_cast<const v4f *>(wstage + q), reinterpret_cast<v4f *>(dst + q));
                        else dst[q] = wstage[q];
                    }
                }
            }
        }
    }
    if (kLds) {
        __syncthreads();
        // flush: one global atomic per touched tile, and the wave's total into the view's entry count (the fill pass derives
        // every tile's list begin from the view totals and the view's own counters: no device-wide scan launch).  The count is
        // spread over kViewSlots words per view: atomics on ONE address are served one after the other at ~75 ns each (measured:
        // 784 per view word made this kernel 87 us instead of 29).
        for (int k = 0; k < nv; ++k) {
            uint32_t *tcg = tile_count + (size_t)(v0 + k) * dm.tiles;
            const uint32_t *hk = hist + (size_t)k * dm.tiles;
            unsigned long long sum = 0;
            for (int t = threadIdx.x; t < dm.tiles; t += 256) {
                const uint32_t h = hk[t];
                if (h) atomicAdd(tcg + t, h);
                sum += h;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
            if (lane == 0 && sum) atomicAdd(view_total + (size_t)(v0 + k) * kViewSlots + ((blockIdx.x * 4 + wave) & (kViewSlots - 1)), sum);
        }
    }
}

void launch_preprocess(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s)
{
    // views per workgroup: as many as keep the LDS histograms within 32 KiB (kLdsTiles words), at most GA_PRE_VIEWS
    static const int max_vg = [] { const char *e = getenv("GA_PRE_VIEWS"); const int v = e ? atoi(e) : kPreViews; return v < 1 ? 1 : v; }();
    static const int nt = [] { const char *e = getenv("GA_PRE_NT"); return e ? atoi(e) : 0; }();
    static const int spt = [] { const char *e = getenv("GA_PRE_SPLATS"); return e ? atoi(e) : kPreSplats; }();
    const bool lds = d.tiles <= kLdsTiles;
    const int vg = std::max(1, std::min(std::min(max_vg, d.V), lds ? kLdsTiles / d.tiles : max_vg));
    const int per = spt >= 4 ? 4 : (spt >= 2 ? 2 : 1);
    const dim3 grid((unsigned)((d.N + 256 * per - 1) / (256 * per)), (unsigned)((d.V + vg - 1) / vg));
#define GA_PRE_LAUNCH(L, S)                                                                                                         \
    hipLaunchKernelGGL((surfel_preprocess_kernel<L, S>), grid, dim3(256), (L) ? (size_t)vg * d.tiles * sizeof(uint32_t) : 0, s,       \
                       a.means3D, a.opacities, a.colors, a.scales, a.rotations, a.viewmatrix, a.projmatrix, a.scale_modifier, d, vg, nt, \
                       a.radii, ws.rect, ws.depth, ws.record, ws.tile_count, ws.view_total)
    if (lds) {
        if (per == 4) GA_PRE_LAUNCH(true, 4); else if (per == 2) GA_PRE_LAUNCH(true, 2); else GA_PRE_LAUNCH(true, 1);
    } else {
        if (per == 4) GA_PRE_LAUNCH(false, 4); else if (per == 2) GA_PRE_LAUNCH(false, 2); else GA_PRE_LAUNCH(false, 1);
    }
#undef GA_PRE_LAUNCH
}

}  // namespace ga
