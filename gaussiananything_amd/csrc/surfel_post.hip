// surfel_post.hip -- the per-pixel post-processing GaussianRenderer2DGS.render applies to the rasterizer's outputs
// (/root/reference/nsr/gs_surfel.py:121-163), for all V views of a batch item in one pass:
//   image       = clamp(color, 0, 1)                                            (:163)
//   rend_normal = (allmap[2:5] as row vectors) @ world_view_transform[:3,:3].T  (:126-128)  view space -> world space
//   depth       = nan_to_num(allmap[5], nan = 0, posinf = 0)                    (:133-134; depth_ratio = 1: median depth)
// alpha (allmap[1]) and dist (allmap[6]) are returned by the host as views of allmap, nothing to compute.
// HBM-bound: 7 planes read, 7 written; one thread per 4 consecutive pixels of a view (16-byte accesses when H*W % 4 == 0).
// (The torch formulation of the rotation alone -- an einsum that lands in a 256x16x16 rocBLAS GEMM -- took 304 us per
// call at 50 views x 512^2 against ~40 us for the whole pass here.)
#include <math.h>

#include "surfel_common.h"

namespace ga {

__device__ __forceinline__ float scrub(float d)
{
    return (isnan(d) || d == INFINITY) ? 0.f : (d == -INFINITY ? -3.4028234663852886e38f : d);  // torch.nan_to_num(x, 0, 0)
}

template <int VEC>
__global__ __launch_bounds__(256) void surfel_postprocess_kernel(const float *__restrict__ color, const float *__restrict__ allmap,
                                                                 const float *__restrict__ viewmatrix, int64_t P,
                                                                 float *__restrict__ image, float *__restrict__ normal,
                                                                 float *__restrict__ depth)
{
    const int v = blockIdx.y;
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p >= P) return;
    const float *vm = viewmatrix + 16 * v;  // row-vector convention: world_view_transform, rows 0..2 = its [:3,:3]
    float R[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[d][c] = vm[4 * d + c];
    const float *col = color + (int64_t)v * 3 * P + p, *am = allmap + (int64_t)v * 7 * P + p;
    float *img = image + (int64_t)v * 3 * P + p, *nrm = normal + (int64_t)v * 3 * P + p, *dep = depth + (int64_t)v * P + p;
    float c[3][VEC], n[3][VEC], d[VEC];
    auto load = [&](const float *src, float (&dst)[VEC]) {
        if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(src);
            dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[VEC - 1] = t.w;
        } else {
            dst[0] = src[0];
        }
    };
    auto store = [&](float *dst, const float (&src)[VEC]) {
        if (VEC == 4) *reinterpret_cast<float4 *>(dst) = make_float4(src[0], src[1], src[2], src[VEC - 1]);
        else dst[0] = src[0];
    };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        load(col + k * P, c[k]);
        load(am + (2 + k) * P, n[k]);
    }
    load(am + 5 * P, d);
    float o[VEC];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = fminf(fmaxf(c[k][e], 0.f), 1.f);
        store(img + k * P, o);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = n[0][e] * R[k][0] + n[1][e] * R[k][1] + n[2][e] * R[k][2];
        store(nrm + k * P, o);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = scrub(d[e]);
    store(dep, o);
}

}  // namespace ga

extern "C" int ga_surfel_postprocess(const GaSurfelPostArgs *a, void *stream)
{
    if (!a || !a->color || !a->allmap || !a->viewmatrix || !a->image || !a->rend_normal || !a->depth) return GA_ERR_NULL_ARG;
    if (a->num_views < 1 || a->num_views > 65535 || a->image_height < 1 || a->image_width < 1) return GA_ERR_BAD_SHAPE;
    const int64_t P = (int64_t)a->image_height * a->image_width;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (P % 4 == 0) {
        const dim3 grid((unsigned)((P / 4 + 255) / 256), (unsigned)a->num_views);
        hipLaunchKernelGGL((ga::surfel_postprocess_kernel<4>), grid, dim3(256), 0, s, a->color, a->allmap, a->viewmatrix, P,
                           a->image, a->rend_normal, a->depth);
    } else {
        const dim3 grid((unsigned)((P + 255) / 256), (unsigned)a->num_views);
        hipLaunchKernelGGL((ga::surfel_postprocess_kernel<1>), grid, dim3(256), 0, s, a->color, a->allmap, a->viewmatrix, P,
                           a->image, a->rend_normal, a->depth);
    }
    return hipGetLastError() == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}
