// surfel_preprocess.hip -- per-(view, Gaussian) preprocess of the 2D-surfel rasterizer, gfx950.
//
// Replaces upstream preprocessCUDA (+ compute_transmat / compute_aabb / getRect) of diff_surfel_rasterization, the
// extension called at /root/reference/nsr/gs_surfel.py:100-114; algorithm per SURVEY.md Appendix A.1 steps 1-7.
//
// This translation unit is built with -ffp-contract=off: radii, tile rects and depth keys must be BIT-IDENTICAL to
// the oracle (oracle/surfel_raster.c), so every expression below keeps the oracle's operation order, one IEEE
// rounding per operation, correctly rounded division and sqrt (hipcc default), no FMA.
//
// MI355X notes: HBM-bound streaming kernel, one thread per (view, Gaussian), view-major so a wave reads 64
// consecutive Gaussians (768 contiguous bytes of means3D).  Outputs are written as ONE 96-byte record per splat
// (what the blend kernel stages into LDS) plus the small SoA side arrays the binning passes stream (depth, rect,
// bbox).  Tile occupancy is counted here (LDS histogram per workgroup, then L2 atomics), which replaces upstream's
// tiles_touched array + device-wide inclusive scan.
#include <hip/hip_fp16.h>

#include "surfel_common.h"

#pragma clang fp contract(off)

namespace ga {

__device__ __forceinline__ int f2i(float f) { return (int)f; }  // v_cvt_i32_f32: toward zero, saturating, NaN -> 0

// One (view, Gaussian).  `tc` is the tile-counter array of this view: the workgroup's LDS histogram (kLds) or the
// global counters.
// Returns true when the record was produced; it is left in `stg` (this lane's 96-byte LDS slot) for the wave's coalesced
// copy-out.
template <bool kLds>
__device__ __forceinline__ bool preprocess_one(
    const float *__restrict__ means3D, const float *__restrict__ opacities, const float *__restrict__ colors,
    const float *__restrict__ scales, const float *__restrict__ rotations, const float *__restrict__ vm,
    const float *__restrict__ pm, float scale_modifier, const Dims &dm, int v, int i, int32_t *__restrict__ radii,
    uint16_t *__restrict__ rect_out, float *__restrict__ depth_out,
    float4 *stg, uint32_t *tc)
{
    const int64_t idx = (int64_t)v * dm.N + i;

    radii[idx] = 0;
    ushort4 rc = make_ushort4(0, 0, 0, 0);
    *reinterpret_cast<ushort4 *>(rect_out + 4 * idx) = rc;

    const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
    const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    if (vz <= 0.2f) return false;

    // A.1 step 2: the quaternion is re-normalised first (upstream quat_to_rotmat); same operation order as the oracle
    const float4 q = *reinterpret_cast<const float4 *>(rotations + 4 * i);
    const float qs = 1.0f / sqrtf(((q.w * q.w + q.x * q.x) + q.y * q.y) + q.z * q.z);
    const float r = q.x * qs, x = q.y * qs, y = q.z * qs, z = q.w * qs;
    const float tu[3] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y + r * z), 2.f * (x * z - r * y)};
    const float tv[3] = {2.f * (x * y - r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + r * x)};
    const float nn[3] = {2.f * (x * z + r * y), 2.f * (y * z - r * x), 1.f - 2.f * (x * x + y * y)};
    const float2 sc = *reinterpret_cast<const float2 *>(scales + 2 * i);
    const float su = scale_modifier * sc.x, sv = scale_modifier * sc.y;

    const float halfW = (float)dm.W / 2.0f, halfH = (float)dm.H / 2.0f;
    const float cW = (float)(dm.W - 1) / 2.0f, cH = (float)(dm.H - 1) / 2.0f;
    const float Hm[3][3] = {{tu[0] * su, tu[1] * su, tu[2] * su}, {tv[0] * sv, tv[1] * sv, tv[2] * sv}, {px, py, pz}};
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

    float nvx = vm[0] * nn[0] + vm[4] * nn[1] + vm[8] * nn[2];
    float nvy = vm[1] * nn[0] + vm[5] * nn[1] + vm[9] * nn[2];
    float nvz = vm[2] * nn[0] + vm[6] * nn[1] + vm[10] * nn[2];
    const float cs = -((vx * nvx + vy * nvy) + vz * nvz);
    if (cs == 0.0f) return false;
    const float mult = cs > 0.0f ? 1.0f : -1.0f;
    nvx = mult * nvx; nvy = mult * nvy; nvz = mult * nvz;

    const float t0 = kCutoff * kCutoff, t1 = kCutoff * kCutoff, t2 = -1.0f;
    const float d = (t0 * (Tw[0] * Tw[0]) + t1 * (Tw[1] * Tw[1])) + t2 * (Tw[2] * Tw[2]);
    if (d == 0.0f) return false;
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
    if ((rmaxx - rminx) * (rmaxy - rminy) == 0) return false;

    radii[idx] = f2i(radius);
    depth_out[idx] = vz;
    rc = make_ushort4((unsigned short)rminx, (unsigned short)rminy, (unsigned short)rmaxx, (unsigned short)rmaxy);
    *reinterpret_cast<ushort4 *>(rect_out + 4 * idx) = rc;

    const float opa = opacities[i];
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
    const float kInf = __builtin_inff();
    float rx = kInf, ry = kInf;
    if (opa < 1.0f / 255.0f) {
        rx = ry = -1.0f;                                 // can never pass the alpha threshold
    } else {
        const float c2 = (2.0f * __logf(255.0f * opa)) * 1.002f + 0.004f;
        const float dd = (c2 * (Tw[0] * Tw[0]) + c2 * (Tw[1] * Tw[1])) - (Tw[2] * Tw[2]);
        if (c2 < 1e30f && dd < 0.0f) {
            const float iv = 1.0f / dd;
            const float g0 = iv * c2, g2 = -iv;
            const float bx = (g0 * (Tu[0] * Tw[0]) + g0 * (Tu[1] * Tw[1])) + g2 * (Tu[2] * Tw[2]);
            const float by = (g0 * (Tv[0] * Tw[0]) + g0 * (Tv[1] * Tw[1])) + g2 * (Tv[2] * Tw[2]);
            const float Ux[3] = {Tu[0] - bx * Tw[0], Tu[1] - bx * Tw[1], Tu[2] - bx * Tw[2]};
            const float Uy[3] = {Tv[0] - by * Tw[0], Tv[1] - by * Tw[1], Tv[2] - by * Tw[2]};
            const float hx = -((g0 * (Ux[0] * Ux[0]) + g0 * (Ux[1] * Ux[1])) + g2 * (Ux[2] * Ux[2]));
            const float hy = -((g0 * (Uy[0] * Uy[0]) + g0 * (Uy[1] * Uy[1])) + g2 * (Uy[2] * Uy[2]));
            const float e3x = sqrtf(fmaxf(hx, 0.0f)) * 1.002f + 0.02f, e3y = sqrtf(fmaxf(hy, 0.0f)) * 1.002f + 0.02f;
            const float r2 = sqrtf(0.5f * c2) * 1.001f + 0.01f;
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
    rec[4] = make_float4(nvx, nvy, nvz, colors[3 * i]);
    rec[5] = make_float4(colors[3 * i + 1], colors[3 * i + 2], 0.0f, 0.0f);

    for (int ty = rminy; ty < rmaxy; ++ty)
        for (int tx = rminx; tx < rmaxx; ++tx) atomicAdd(tc + ty * dm.gx + tx, 1u);
    return true;
}

// One workgroup = 256 threads x kPreSplats consecutive Gaussians of ONE view (blockIdx.y).  Tile occupancy is
// accumulated in an LDS histogram of the view's tiles and flushed with one global atomic per touched tile per
// workgroup: the hottest tile of a real scene receives thousands of increments per view and same-address L2 atomics
// serialise (measured 0.27 ms for 1.4 M increments, profiles/r1a_*).  Views with more than kLdsTiles tiles use the
// global counters directly.
template <bool kLds>
__global__ __launch_bounds__(256) void surfel_preprocess_kernel(
    const float *__restrict__ means3D, const float *__restrict__ opacities, const float *__restrict__ colors,
    const float *__restrict__ scales, const float *__restrict__ rotations, const float *__restrict__ viewmatrix,
    const float *__restrict__ projmatrix, float scale_modifier, Dims dm, int32_t *__restrict__ radii,
    uint16_t *__restrict__ rect_out, float *__restrict__ depth_out,
    float *__restrict__ rec_out, uint32_t *__restrict__ tile_count, unsigned long long *__restrict__ view_total)
{
    extern __shared__ uint32_t hist[];
    // Records leave through LDS: a lane's record is 96 contiguous bytes, so direct stores would be six 16-byte pieces
    // at a 96-byte stride per instruction (partial lines); the wave's 64 records are one contiguous 6 KiB block, written
    // with six fully coalesced 1 KiB stores instead.
    __shared__ __attribute__((aligned(16))) float4 stage[256 * (kRec / 4)];
    const int v = blockIdx.y;
    if (kLds) {
        for (int t = threadIdx.x; t < dm.tiles; t += 256) hist[t] = 0;
        __syncthreads();
    }
    const float *vm = viewmatrix + 16 * v, *pm = projmatrix + 16 * v;
    uint32_t *tcg = tile_count + (size_t)v * dm.tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *wstage = stage + wave * 64 * (kRec / 4);
    for (int k = 0; k < kPreSplats; ++k) {
        const int i = (blockIdx.x * kPreSplats + k) * 256 + threadIdx.x;
        bool live = false;
        if (i < dm.N)
            live = preprocess_one<kLds>(means3D, opacities, colors, scales, rotations, vm, pm, scale_modifier, dm, v, i, radii,
                                        rect_out, depth_out, wstage + lane * (kRec / 4), kLds ? hist : tcg);
        if (__builtin_amdgcn_ballot_w64(live) != 0) {  // wave-uniform; records of culled lanes are never read
            const int first = i - lane;                                    // first Gaussian of this wave
            const int nq = min(64, dm.N - first) * (kRec / 4);             // float4s inside the array
            float4 *dst = reinterpret_cast<float4 *>(rec_out + ((size_t)v * dm.N + first) * kRec);
#pragma unroll
            for (int j = 0; j < kRec / 4; ++j) {
                const int q = lane + 64 * j;
                if (q < nq) dst[q] = wstage[q];
            }
        }
    }
    if (kLds) {
        __syncthreads();
        // flush: one global atomic per touched tile, and the wave's total into the view's entry count (the fill pass derives
        // every tile's list begin from the view totals and the view's own counters: no device-wide scan launch).  The count is
        // spread over kViewSlots words per view: atomics on ONE address are served one after the other at ~75 ns each (measured:
        // 784 per view word made this kernel 87 us instead of 29).
        unsigned long long mine = 0;
        for (int t = threadIdx.x; t < dm.tiles; t += 256) {
            const uint32_t h = hist[t];
            if (h) atomicAdd(tcg + t, h);
            mine += h;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
        if (lane == 0 && mine) atomicAdd(view_total + (size_t)v * kViewSlots + ((blockIdx.x * 4 + wave) & (kViewSlots - 1)), mine);
    }
}

void launch_preprocess(const GaSurfelForwardArgs &a, const Dims &d, const Workspace &ws, hipStream_t s)
{
    const dim3 grid((unsigned)((d.N + 256 * kPreSplats - 1) / (256 * kPreSplats)), (unsigned)d.V);
    if (d.tiles <= kLdsTiles)
        hipLaunchKernelGGL(surfel_preprocess_kernel<true>, grid, dim3(256), d.tiles * sizeof(uint32_t), s, a.means3D,
                           a.opacities, a.colors, a.scales, a.rotations, a.viewmatrix, a.projmatrix, a.scale_modifier,
                           d, a.radii, ws.rect, ws.depth, ws.record, ws.tile_count, ws.view_total);
    else
        hipLaunchKernelGGL(surfel_preprocess_kernel<false>, grid, dim3(256), 0, s, a.means3D, a.opacities, a.colors,
                           a.scales, a.rotations, a.viewmatrix, a.projmatrix, a.scale_modifier, d, a.radii, ws.rect,
                           ws.depth, ws.record, ws.tile_count, ws.view_total);
}

}  // namespace ga
