// tsdf.hip -- TSDF fusion of rendered RGB-D frames and marching-cubes mesh extraction, gfx950.
//
// What the reference does on the CPU with Open3D (third party, absent here) for the mesh export of a generated object:
// /root/reference/nsr/lsgm/flow_matching_trainer.py:1318-1395 (extract_mesh_bounded: ScalableTSDFVolume.integrate per camera,
// extract_triangle_mesh).  The arithmetic follows Open3D's published implementation as restated in oracle/tsdf.py
// (UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier, ScalableTSDFVolume::Integrate / ExtractTriangleMesh);
// the structure does not: the hash map of 16^3-voxel units is a dense array of units over the object's bounding cube
// (include/ga_tsdf.h), a frame is two launches (open units, integrate opened units: one workgroup per unit, one thread per
// (x, y) column walking z exactly as Open3D's loop does, unit-blocked storage so that the 256 columns of a z-slice are one
// contiguous kilobyte), and the mesh is four launches over the allocated units with device-side counts and offsets.
// HBM-bound streaming work: 20 B per voxel per frame for the opened units.  Compiled with -ffp-contract=off (the float
// expressions are Open3D's, operation by operation).
#include <hip/hip_runtime.h>

#include "../../include/ga_tsdf.h"
#include "mc_table.h"

namespace ga {

constexpr int kUnitVox = GA_TSDF_UNIT * GA_TSDF_UNIT * GA_TSDF_UNIT;

struct VolP {
    int Ux, Uy, Uz, u0x, u0y, u0z;
    double voxel_length, unit_length, sdf_trunc;
    float *tsdf, *weight, *color;
    uint8_t *touched, *allocated;
    size_t nvox;
};

struct FrameP {
    int H, W, stride;
    const float *rgb, *depth, *alpha;
    float alpha_thres, depth_trunc;
    double fx, fy, cx, cy;
    double pose[16];
    float ext[16];
};

__device__ __forceinline__ size_t vox_addr(const VolP &v, int gx, int gy, int gz)
{
    const int ux = gx >> 4, uy = gy >> 4, uz = gz >> 4;
    return (size_t)((ux * v.Uy + uy) * v.Uz + uz) * kUnitVox + ((gz & 15) << 8) + ((gx & 15) << 4) + (gy & 15);
}

// the depth Open3D's RGBDImage holds: zero where the reference masks it (alpha below the threshold) or at / beyond depth_trunc
__device__ __forceinline__ float frame_depth(const FrameP &f, int u, int v)
{
    const size_t p = (size_t)v * f.W + u;
    float d = f.depth[p];
    if (f.alpha && f.alpha[p] < f.alpha_thres) d = 0.0f;
    if (d >= f.depth_trunc) d = 0.0f;
    return d;
}

// ScalableTSDFVolume::Integrate, first half: the units within sdf_trunc of every depth_sampling_stride-th depth point
__global__ __launch_bounds__(256) void tsdf_open_units_kernel(VolP vol, FrameP f)
{
    const int sw = (f.W + f.stride - 1) / f.stride, sh = (f.H + f.stride - 1) / f.stride;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= sw * sh) return;
    const int i = (t / sw) * f.stride, j = (t % sw) * f.stride;
    const float d = frame_depth(f, j, i);
    if (!(d > 0.0f)) return;
    const double z = (double)d, x = ((double)j - f.cx) * z / f.fx, y = ((double)i - f.cy) * z / f.fy;
    double p[3];
    for (int r = 0; r < 3; ++r) p[r] = ((f.pose[4 * r] * x + f.pose[4 * r + 1] * y) + f.pose[4 * r + 2] * z) + f.pose[4 * r + 3];
    int lo[3], hi[3];
    const int u0[3] = {vol.u0x, vol.u0y, vol.u0z}, U[3] = {vol.Ux, vol.Uy, vol.Uz};
    for (int r = 0; r < 3; ++r) {
        lo[r] = (int)floor((p[r] - vol.sdf_trunc) / vol.unit_length) - u0[r];
        hi[r] = (int)floor((p[r] + vol.sdf_trunc) / vol.unit_length) - u0[r];
        lo[r] = max(lo[r], 0);
        hi[r] = min(hi[r], U[r] - 1);   // (units outside the box are not allocated)
    }
    for (int a = lo[0]; a <= hi[0]; ++a)
        for (int b = lo[1]; b <= hi[1]; ++b)
            for (int c = lo[2]; c <= hi[2]; ++c) {
                const int u = (a * vol.Uy + b) * vol.Uz + c;
                vol.touched[u] = 1;
                vol.allocated[u] = 1;
            }
}

// UniformTSDFVolume::IntegrateWithDepthToCameraDistanceMultiplier over the voxels of one opened unit
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(VolP vol, FrameP f)
{
    const int u = blockIdx.x;
    if (!vol.touched[u]) return;
    const int uz = u % vol.Uz, uy = (u / vol.Uz) % vol.Uy, ux = u / (vol.Uz * vol.Uy);
    const int lx = threadIdx.x >> 4, ly = threadIdx.x & 15;
    const float vl = (float)vol.voxel_length, half = vl * 0.5f;
    const float trunc = (float)vol.sdf_trunc, trunc_inv = 1.0f / trunc;
    const float fx = (float)f.fx, fy = (float)f.fy, cx = (float)f.cx, cy = (float)f.cy;
    const float safe_w = (float)f.W - 0.0001f, safe_h = (float)f.H - 0.0001f;
    const float ffl_inv0 = 1.0f / fx, ffl_inv1 = 1.0f / fy;
    const double ox = (double)(vol.u0x + ux) * vol.unit_length, oy = (double)(vol.u0y + uy) * vol.unit_length,
                 oz = (double)(vol.u0z + uz) * vol.unit_length;
    const float px = (float)((double)(half + vl * (float)lx) + ox), py = (float)((double)(half + vl * (float)ly) + oy),
                pz = (float)((double)half + oz);
    float pc[3], step[3];
    for (int r = 0; r < 3; ++r) {
        pc[r] = ((f.ext[4 * r] * px + f.ext[4 * r + 1] * py) + f.ext[4 * r + 2] * pz) + f.ext[4 * r + 3];
        step[r] = f.ext[4 * r + 2] * vl;
    }
    const size_t base = (size_t)u * kUnitVox + threadIdx.x;   // (lx * 16 + ly = threadIdx.x)
    const size_t HW = (size_t)f.H * f.W;
    for (int lz = 0; lz < GA_TSDF_UNIT; ++lz, pc[0] += step[0], pc[1] += step[1], pc[2] += step[2]) {
        if (!(pc[2] > 0.0f)) continue;
        const float u_f = pc[0] * fx / pc[2] + cx + 0.5f, v_f = pc[1] * fy / pc[2] + cy + 0.5f;
        if (!(u_f >= 0.0001f && u_f < safe_w && v_f >= 0.0001f && v_f < safe_h)) continue;
        const int iu = (int)u_f, iv = (int)v_f;
        const float d = frame_depth(f, iu, iv);
        if (!(d > 0.0f)) continue;
        const float xx = ((float)iu - cx) * ffl_inv0, yy = ((float)iv - cy) * ffl_inv1;
        const float mult = sqrtf(xx * xx + yy * yy + 1.0f);
        const float sdf = (d - pc[2]) * mult;
        if (!(sdf > -trunc)) continue;
        const size_t a = base + ((size_t)lz << 8);
        const float tsdf = fminf(1.0f, sdf * trunc_inv), w = vol.weight[a], w1 = w + 1.0f;
        vol.tsdf[a] = (vol.tsdf[a] * w + tsdf) / w1;
        const size_t pix = (size_t)iv * f.W + iu;
        for (int c = 0; c < 3; ++c) {
            const float v01 = fminf(fmaxf(f.rgb[c * HW + pix], 0.0f), 1.0f);
            const float c8 = (float)(unsigned char)(v01 * 255.0f);   // (np.asarray(clip(rgb) * 255, dtype=uint8): truncation)
            float *cp = vol.color + (size_t)c * vol.nvox + a;
            *cp = (*cp * w + c8) / w1;
        }
        vol.weight[a] = w1;
    }
}

// ---- mesh extraction ----------------------------------------------------------------------------------------------------------
struct MeshP {
    uint8_t *cube;        // [nvox] marching-cubes case of the cube anchored at the voxel (0: not all 8 corners observed, or empty)
    uint8_t *vflags;      // [nvox] bit a: the edge from the voxel along axis a carries a vertex
    int32_t *vid;         // [nvox] index of the voxel's first vertex
    int32_t *unit_count;  // [units][2] vertices, triangles of the unit
    int32_t *unit_base;   // [units][2] exclusive prefix
};

__device__ __forceinline__ void unit_coords(const VolP &v, int u, int &ux, int &uy, int &uz)
{
    uz = u % v.Uz; uy = (u / v.Uz) % v.Uy; ux = u / (v.Uz * v.Uy);
}

// thread t of a unit's workgroup owns the 16 voxels t * 16 .. t * 16 + 15 in storage order: lz = t >> 4, lx = t & 15, ly = 0..15
__global__ __launch_bounds__(256) void mc_classify_kernel(VolP vol, MeshP m)
{
    const int u = blockIdx.x;
    const size_t base = (size_t)u * kUnitVox + (size_t)threadIdx.x * 16;
    if (!vol.allocated[u]) {
        for (int k = 0; k < 16; ++k) m.cube[base + k] = 0;
        return;
    }
    int ux, uy, uz;
    unit_coords(vol, u, ux, uy, uz);
    const int gx = ux * 16 + (threadIdx.x & 15), gz = uz * 16 + (threadIdx.x >> 4);
    const int Rx = vol.Ux * 16, Ry = vol.Uy * 16, Rz = vol.Uz * 16;
    for (int ly = 0; ly < 16; ++ly) {
        const int gy = uy * 16 + ly;
        int cs = 0;
        bool valid = gx + 1 < Rx && gy + 1 < Ry && gz + 1 < Rz;
        for (int i = 0; i < 8 && valid; ++i) {
            const size_t a = vox_addr(vol, gx + (i & 1), gy + ((i >> 1) & 1), gz + (i >> 2));
            if (vol.weight[a] == 0.0f) valid = false;
            else if (vol.tsdf[a] < 0.0f) cs |= 1 << i;
        }
        m.cube[base + ly] = valid && cs != 255 ? (uint8_t)cs : (uint8_t)0;
    }
}

__device__ __forceinline__ int cube_at(const VolP &vol, const MeshP &m, int gx, int gy, int gz)
{
    if (gx < 0 || gy < 0 || gz < 0) return 0;
    return m.cube[vox_addr(vol, gx, gy, gz)];
}
__device__ __forceinline__ int bits_differ(int cs, int a, int b) { return ((cs >> a) ^ (cs >> b)) & 1; }

// the edges owned by voxel p that carry a vertex: intersected in one of the (up to four) valid cubes around the edge
__device__ __forceinline__ int vertex_flags(const VolP &vol, const MeshP &m, int gx, int gy, int gz)
{
    const int c = cube_at(vol, m, gx, gy, gz), cx = cube_at(vol, m, gx - 1, gy, gz), cy = cube_at(vol, m, gx, gy - 1, gz),
              cz = cube_at(vol, m, gx, gy, gz - 1), cxy = cube_at(vol, m, gx - 1, gy - 1, gz),
              cxz = cube_at(vol, m, gx - 1, gy, gz - 1), cyz = cube_at(vol, m, gx, gy - 1, gz - 1);
    const int fx = bits_differ(c, 0, 1) | bits_differ(cy, 2, 3) | bits_differ(cz, 4, 5) | bits_differ(cyz, 6, 7);
    const int fy = bits_differ(c, 0, 2) | bits_differ(cx, 1, 3) | bits_differ(cz, 4, 6) | bits_differ(cxz, 5, 7);
    const int fz = bits_differ(c, 0, 4) | bits_differ(cx, 1, 5) | bits_differ(cy, 2, 6) | bits_differ(cxy, 3, 7);
    return fx | (fy << 1) | (fz << 2);
}

// exclusive prefix of one int per thread over the 256 threads of the workgroup; `total` receives the sum
__device__ __forceinline__ int block_exclusive(int v, int *sh4, int &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) sh4[w] = inc;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < w; ++k) off += sh4[k];
    total = sh4[0] + sh4[1] + sh4[2] + sh4[3];
    return off + inc - v;
}

__global__ __launch_bounds__(256) void mc_count_kernel(VolP vol, MeshP m)
{
    __shared__ int sh4[4];
    const int u = blockIdx.x;
    const size_t base = (size_t)u * kUnitVox + (size_t)threadIdx.x * 16;
    int nv = 0, nt = 0;
    if (vol.allocated[u]) {   // (a voxel of a unit never opened has weight 0: it owns no vertex and anchors no valid cube)
        int ux, uy, uz;
        unit_coords(vol, u, ux, uy, uz);
        const int gx = ux * 16 + (threadIdx.x & 15), gz = uz * 16 + (threadIdx.x >> 4);
        for (int ly = 0; ly < 16; ++ly) {
            const int fl = vertex_flags(vol, m, gx, uy * 16 + ly, gz);
            m.vflags[base + ly] = (uint8_t)fl;
            nv += __popc(fl);
            nt += kMcTriangleCount[m.cube[base + ly]];
        }
    } else {
        for (int k = 0; k < 16; ++k) m.vflags[base + k] = 0;
    }
    int tv, tt;
    block_exclusive(nv, sh4, tv);
    block_exclusive(nt, sh4, tt);
    if (threadIdx.x == 0) { m.unit_count[2 * u] = tv; m.unit_count[2 * u + 1] = tt; }
}

__global__ __launch_bounds__(1024) void mc_scan_units_kernel(MeshP m, int units, int64_t *counts)
{
    __shared__ long long wave_tot[2][16];
    __shared__ long long carry[2];
    if (threadIdx.x < 2) carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int b = 0; b < units; b += 1024) {
        const int u = b + (int)threadIdx.x;
        long long v[2] = {0, 0}, inc[2];
        if (u < units) { v[0] = m.unit_count[2 * u]; v[1] = m.unit_count[2 * u + 1]; }
        for (int k = 0; k < 2; ++k) {
            inc[k] = v[k];
            for (int d = 1; d < 64; d <<= 1) {
                const long long o = __shfl_up(inc[k], d, 64);
                if (lane >= d) inc[k] += o;
            }
            if (lane == 63) wave_tot[k][w] = inc[k];
        }
        __syncthreads();
        long long off[2] = {carry[0], carry[1]};
        for (int k = 0; k < 2; ++k)
            for (int q = 0; q < w; ++q) off[k] += wave_tot[k][q];
        if (u < units) {
            m.unit_base[2 * u] = (int32_t)(off[0] + inc[0] - v[0]);
            m.unit_base[2 * u + 1] = (int32_t)(off[1] + inc[1] - v[1]);
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry[0] = off[0] + inc[0]; carry[1] = off[1] + inc[1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { counts[0] = carry[0]; counts[1] = carry[1]; }
}

// ExtractTriangleMesh's vertex: on the edge from voxel p along `axis`, at |f0| / (|f0| + |f1|) of a voxel from p; colour likewise
__global__ __launch_bounds__(256) void mc_vertices_kernel(VolP vol, MeshP m, float *__restrict__ vertices, float *__restrict__ colors,
                                                          int cap)
{
    __shared__ int sh4[4];
    const int u = blockIdx.x;
    if (m.unit_count[2 * u] == 0) return;
    const size_t base = (size_t)u * kUnitVox + (size_t)threadIdx.x * 16;
    int nv = 0;
    for (int k = 0; k < 16; ++k) nv += __popc(m.vflags[base + k]);
    int total;
    int id = m.unit_base[2 * u] + block_exclusive(nv, sh4, total);
    int ux, uy, uz;
    unit_coords(vol, u, ux, uy, uz);
    const int lx = threadIdx.x & 15, lz = threadIdx.x >> 4;
    const double vl = vol.voxel_length, half = vl * 0.5;
    const double org[3] = {(double)(vol.u0x + ux) * vol.unit_length, (double)(vol.u0y + uy) * vol.unit_length,
                           (double)(vol.u0z + uz) * vol.unit_length};
    for (int ly = 0; ly < 16; ++ly) {
        const int fl = m.vflags[base + ly];
        m.vid[base + ly] = id;
        if (!fl) continue;
        const int g[3] = {ux * 16 + lx, uy * 16 + ly, uz * 16 + lz}, l[3] = {lx, ly, lz};
        const size_t a0 = base + ly;
        const double f0 = fabs((double)vol.tsdf[a0]);
        for (int axis = 0; axis < 3; ++axis) {
            if (!((fl >> axis) & 1)) continue;
            const size_t a1 = vox_addr(vol, g[0] + (axis == 0), g[1] + (axis == 1), g[2] + (axis == 2));
            const double f1 = fabs((double)vol.tsdf[a1]);
            double pt[3] = {half + vl * l[0], half + vl * l[1], half + vl * l[2]};
            pt[axis] += f0 * vl / (f0 + f1);
            for (int c = 0; c < 3 && id < cap; ++c) {
                vertices[3 * (size_t)id + c] = (float)(pt[c] + org[c]);
                const double c0 = (double)vol.color[(size_t)c * vol.nvox + a0], c1 = (double)vol.color[(size_t)c * vol.nvox + a1];
                colors[3 * (size_t)id + c] = (float)(((f1 * c0 + f0 * c1) / (f0 + f1)) / 255.0);
            }
            ++id;
        }
    }
}

__global__ __launch_bounds__(256) void mc_triangles_kernel(VolP vol, MeshP m, int32_t *__restrict__ triangles, int cap)
{
    __shared__ int sh4[4];
    const int u = blockIdx.x;
    if (m.unit_count[2 * u + 1] == 0) return;
    const size_t base = (size_t)u * kUnitVox + (size_t)threadIdx.x * 16;
    int nt = 0;
    for (int k = 0; k < 16; ++k) nt += kMcTriangleCount[m.cube[base + k]];
    int total;
    int tri = m.unit_base[2 * u + 1] + block_exclusive(nt, sh4, total);
    int ux, uy, uz;
    unit_coords(vol, u, ux, uy, uz);
    const int gx = ux * 16 + (threadIdx.x & 15), gz = uz * 16 + (threadIdx.x >> 4);
    for (int ly = 0; ly < 16; ++ly) {
        const int cs = m.cube[base + ly], n = kMcTriangleCount[cs];
        const int gy = uy * 16 + ly;
        for (int k = 0; k < 3 * n && tri + n <= cap; ++k) {
            // edge e = 4 axis + 2 b + a: its owner is the voxel at the edge's lower corner
            const int e = kMcTriangles[cs][k], axis = e >> 2, a = e & 1, b = (e >> 1) & 1;
            const int qx = gx + (axis == 0 ? 0 : a), qy = gy + (axis == 0 ? a : (axis == 1 ? 0 : b)), qz = gz + (axis == 2 ? 0 : b);
            const size_t q = vox_addr(vol, qx, qy, qz);
            triangles[3 * (size_t)tri + k] = m.vid[q] + __popc(m.vflags[q] & ((1 << axis) - 1));
        }
        tri += n;
    }
}

static bool make_vol(const GaTsdfVolume *v, VolP &o)
{
    if (!v) return false;
    const int64_t units = (int64_t)v->units[0] * v->units[1] * v->units[2];
    if (v->units[0] <= 0 || v->units[1] <= 0 || v->units[2] <= 0 || units > (1 << 24) || !(v->voxel_length > 0.0) || !(v->sdf_trunc > 0.0))
        return false;
    o.Ux = v->units[0]; o.Uy = v->units[1]; o.Uz = v->units[2];
    o.u0x = v->unit0[0]; o.u0y = v->unit0[1]; o.u0z = v->unit0[2];
    o.voxel_length = v->voxel_length; o.unit_length = v->voxel_length * GA_TSDF_UNIT; o.sdf_trunc = v->sdf_trunc;
    o.tsdf = v->tsdf; o.weight = v->weight; o.color = v->color; o.touched = v->touched; o.allocated = v->allocated;
    o.nvox = (size_t)units * kUnitVox;
    return true;
}

struct MeshScratch { size_t cube, vflags, vid, unit_count, unit_base, total; };

static void mesh_layout(const VolP &v, MeshScratch &o)
{
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t units = (size_t)v.Ux * v.Uy * v.Uz;
    o.cube = take(v.nvox); o.vflags = take(v.nvox); o.vid = take(v.nvox * 4);
    o.unit_count = take(units * 8); o.unit_base = take(units * 8);
    o.total = off;
}

static MeshP make_mesh(void *scratch, const MeshScratch &s)
{
    unsigned char *p = static_cast<unsigned char *>(scratch);
    MeshP m;
    m.cube = p + s.cube; m.vflags = p + s.vflags; m.vid = reinterpret_cast<int32_t *>(p + s.vid);
    m.unit_count = reinterpret_cast<int32_t *>(p + s.unit_count); m.unit_base = reinterpret_cast<int32_t *>(p + s.unit_base);
    return m;
}

}  // namespace ga

extern "C" int ga_tsdf_integrate(const GaTsdfVolume *volume, const GaTsdfFrame *fr, void *stream_v)
{
    using namespace ga;
    if (!volume || !fr) return GA_ERR_NULL_ARG;
    VolP v;
    if (!make_vol(volume, v)) return GA_ERR_BAD_SHAPE;
    if (!v.tsdf || !v.weight || !v.color || !v.touched || !v.allocated || !fr->rgb || !fr->depth) return GA_ERR_NULL_ARG;
    if (fr->height <= 0 || fr->width <= 0 || fr->depth_sampling_stride <= 0 || !(fr->fx > 0.0) || !(fr->fy > 0.0)) return GA_ERR_BAD_SHAPE;
    FrameP f;
    f.H = fr->height; f.W = fr->width; f.stride = fr->depth_sampling_stride;
    f.rgb = fr->rgb; f.depth = fr->depth; f.alpha = fr->alpha; f.alpha_thres = fr->alpha_thres; f.depth_trunc = fr->depth_trunc;
    f.fx = fr->fx; f.fy = fr->fy; f.cx = fr->cx; f.cy = fr->cy;
    for (int k = 0; k < 16; ++k) { f.pose[k] = fr->pose[k]; f.ext[k] = (float)fr->extrinsic[k]; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_v);
    (void)hipGetLastError();
    const unsigned units = (unsigned)(v.Ux * v.Uy * v.Uz);
    (void)hipMemsetAsync(v.touched, 0, units, s);
    const int samples = ((f.W + f.stride - 1) / f.stride) * ((f.H + f.stride - 1) / f.stride);
    hipLaunchKernelGGL(tsdf_open_units_kernel, dim3((unsigned)((samples + 255) / 256)), dim3(256), 0, s, v, f);
    hipLaunchKernelGGL(tsdf_integrate_kernel, dim3(units), dim3(256), 0, s, v, f);
    return hipGetLastError() == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

extern "C" size_t ga_tsdf_mesh_scratch_bytes(const GaTsdfVolume *volume)
{
    ga::VolP v;
    if (!ga::make_vol(volume, v)) return 0;
    ga::MeshScratch s;
    ga::mesh_layout(v, s);
    return s.total;
}

extern "C" int ga_tsdf_mesh_count(const GaTsdfVolume *volume, void *scratch, size_t scratch_bytes, int64_t *counts, void *stream_v)
{
    using namespace ga;
    VolP v;
    if (!make_vol(volume, v)) return GA_ERR_BAD_SHAPE;
    if (!scratch || !counts || !v.tsdf || !v.weight || !v.color || !v.allocated) return GA_ERR_NULL_ARG;
    MeshScratch ms;
    mesh_layout(v, ms);
    if (scratch_bytes < ms.total) return GA_ERR_WORKSPACE;
    const MeshP m = make_mesh(scratch, ms);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_v);
    (void)hipGetLastError();
    const unsigned units = (unsigned)(v.Ux * v.Uy * v.Uz);
    hipLaunchKernelGGL(mc_classify_kernel, dim3(units), dim3(256), 0, s, v, m);
    hipLaunchKernelGGL(mc_count_kernel, dim3(units), dim3(256), 0, s, v, m);
    hipLaunchKernelGGL(mc_scan_units_kernel, dim3(1), dim3(1024), 0, s, m, (int)units, counts);
    return hipGetLastError() == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

extern "C" int ga_tsdf_mesh_emit(const GaTsdfVolume *volume, void *scratch, size_t scratch_bytes, int64_t num_vertices,
                                 int64_t num_triangles, float *vertices, float *colors, int32_t *triangles, void *stream_v)
{
    using namespace ga;
    VolP v;
    if (!make_vol(volume, v)) return GA_ERR_BAD_SHAPE;
    if (!scratch) return GA_ERR_NULL_ARG;
    if (num_vertices < 0 || num_triangles < 0 || num_vertices > 0x7fffffffll / 3 || num_triangles > 0x7fffffffll / 3) return GA_ERR_BAD_SHAPE;
    if ((num_vertices > 0 && (!vertices || !colors)) || (num_triangles > 0 && !triangles)) return GA_ERR_NULL_ARG;
    MeshScratch ms;
    mesh_layout(v, ms);
    if (scratch_bytes < ms.total) return GA_ERR_WORKSPACE;
    const MeshP m = make_mesh(scratch, ms);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_v);
    (void)hipGetLastError();
    const unsigned units = (unsigned)(v.Ux * v.Uy * v.Uz);
    if (num_vertices > 0) hipLaunchKernelGGL(mc_vertices_kernel, dim3(units), dim3(256), 0, s, v, m, vertices, colors, (int)num_vertices);
    if (num_triangles > 0) hipLaunchKernelGGL(mc_triangles_kernel, dim3(units), dim3(256), 0, s, v, m, triangles, (int)num_triangles);
    return hipGetLastError() == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

// host: Wavefront OBJ with per-vertex colours (`v x y z r g b`, `f a b c` one-based) -- what o3d.io.write_triangle_mesh writes
// for the reference (flow_matching_trainer.py:1297, 1311).  Host arrays; returns 0, or GA_ERR_NULL_ARG / GA_ERR_LAUNCH (I/O).
#include <cstdio>
#include <vector>
// ---- connected triangle clusters (utils/mesh_util.py:22-44 -> Open3D cluster_connected_triangles: triangles joined through shared
// edges).  Lock-free union-find over the pairs of triangles that share an edge: a pair hooks the LARGER of its two roots under the
// smaller one with a compare-and-swap and retries until both ends have one root, so one pass over the pairs is complete and the root of
// a cluster is its smallest triangle index whatever the order the pairs are served in (deterministic labels).  Round 4 did this with
// device-wide torch operations (min-label propagation + pointer jumping, seven rounds of ~8 launches over 1 M triangles: 9.9 ms).
namespace gamesh {

__device__ __forceinline__ int32_t cc_find(int32_t *parent, int32_t x)
{
    int32_t p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) {   // path halving (parents only ever decrease: a stale read costs a step, never correctness)
        const int32_t g = __hip_atomic_load(parent + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (g != p) __hip_atomic_store(parent + x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = g;
    }
    return x;
}

__global__ __launch_bounds__(256) void cc_init_kernel(int32_t *parent, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) parent[i] = (int32_t)i;
}

__global__ __launch_bounds__(256) void cc_hook_kernel(const int64_t *__restrict__ a, const int64_t *__restrict__ b, int64_t npairs, int32_t *parent)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    int32_t ra = cc_find(parent, (int32_t)a[i]), rb = cc_find(parent, (int32_t)b[i]);
    while (ra != rb) {
        if (ra < rb) { const int32_t t = ra; ra = rb; rb = t; }     // ra > rb: hook ra under rb if ra is still a root
        const int32_t prev = atomicCAS(parent + ra, ra, rb);
        if (prev == ra) break;
        ra = cc_find(parent, prev);                                   // somebody hooked it first: follow and retry
        rb = cc_find(parent, rb);
    }
}

__global__ __launch_bounds__(256) void cc_flatten_kernel(int32_t *parent, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        // read-only walk: a path-halving store of another thread could land AFTER this thread's final store and leave a non-root label
        int32_t x = (int32_t)i, p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (p != x) { x = p; p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        __hip_atomic_store(parent + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (readers see the old parent or the root: both lead to the root)
    }
}

}  // namespace gamesh

extern "C" int ga_mesh_cluster_labels(const int64_t *pair_a, const int64_t *pair_b, int64_t num_pairs, int32_t *labels, int64_t num_triangles,
                                      void *stream_v)
{
    using namespace gamesh;
    if (!labels || (num_pairs > 0 && (!pair_a || !pair_b))) return GA_ERR_NULL_ARG;
    if (num_triangles <= 0 || num_triangles > 0x7FFFFFFFll || num_pairs < 0) return GA_ERR_BAD_SHAPE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream_v);
    (void)hipGetLastError();
    hipLaunchKernelGGL(cc_init_kernel, dim3((unsigned)((num_triangles + 255) / 256)), dim3(256), 0, s, labels, num_triangles);
    if (num_pairs > 0)
        hipLaunchKernelGGL(cc_hook_kernel, dim3((unsigned)((num_pairs + 255) / 256)), dim3(256), 0, s, pair_a, pair_b, num_pairs, labels);
    hipLaunchKernelGGL(cc_flatten_kernel, dim3((unsigned)((num_triangles + 255) / 256)), dim3(256), 0, s, labels, num_triangles);
    return hipGetLastError() == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

extern "C" int ga_mesh_write_obj(const char *path, const float *vertices, const float *colors, const int32_t *triangles,
                                 int64_t num_vertices, int64_t num_triangles)
{
    if (!path || (num_vertices > 0 && !vertices) || (num_triangles > 0 && !triangles) || num_vertices < 0 || num_triangles < 0)
        return GA_ERR_NULL_ARG;
    FILE *f = std::fopen(path, "w");
    if (!f) return GA_ERR_LAUNCH;
    std::vector<char> buf(1 << 22);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    std::fputs("# GaussianAnything mesh export (TSDF fusion of the rendered views)\n", f);
    for (int64_t i = 0; i < num_vertices; ++i) {
        const float *v = vertices + 3 * i;
        if (colors) {
            const float *c = colors + 3 * i;
            std::fprintf(f, "v %.6f %.6f %.6f %.6f %.6f %.6f\n", v[0], v[1], v[2], c[0], c[1], c[2]);
        } else {
            std::fprintf(f, "v %.6f %.6f %.6f\n", v[0], v[1], v[2]);
        }
    }
    for (int64_t i = 0; i < num_triangles; ++i) {
        const int32_t *t = triangles + 3 * i;
        std::fprintf(f, "f %d %d %d\n", t[0] + 1, t[1] + 1, t[2] + 1);
    }
    const bool bad = std::ferror(f) != 0;
    return (std::fclose(f) != 0 || bad) ? GA_ERR_LAUNCH : GA_OK;
}
