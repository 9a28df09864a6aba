/*
 * oracle/surfel_raster.c -- CPU restatement of the 2D-surfel ("2DGS") tile rasterizer forward.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gaussiananything_amd/ may import, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED.  The arithmetic restated here lives in the third-party CUDA extension
 * `diff_surfel_rasterization` (git+https://github.com/hbb1/diff-surfel-rasterization.git, no pinned
 * commit: /root/reference/README.md:158-159; submodule entry commented out, /root/reference/.gitmodules:1-6),
 * which is absent from /root/reference and cannot be built here (CUDA only).  The reference holds no
 * tests or golden vectors for it.  This file restates the published algorithm
 * (cuda_rasterizer/{auxiliary.h,forward.cu,rasterizer_impl.cu,config.h} of that project, as summarised
 * in SURVEY.md Appendix A.1) and is anchored on the reference's own call site and consumer:
 *   - call convention and argument meaning : /root/reference/nsr/gs_surfel.py:85-114
 *   - allmap channel meaning               : /root/reference/nsr/gs_surfel.py:121-142
 *   - row-vector matrix convention         : /root/reference/nsr/lsgm/flow_matching_trainer.py:2196-2205
 * Short of the unavailable source, the one external anchor is the published method itself: tests/test_cpu_oracle_and_host.py
 * (test_surfel_oracle_against_the_published_method) renders a fronto-parallel, a tilted and three overlapping surfels with an
 * independent float64 ray-splat solve written from the 2DGS paper's equations (no homography trick, no tiles) and this file
 * agrees with it to 5e-5 on colour, alpha, normals, expected / median depth and distortion.  That pins the method, not
 * upstream's implementation choices (cut-offs, tile culling, tie order), which remain as recalled in SURVEY.md A.1.  One
 * such choice is visible against the paper: the screen-space low-pass filter is centred on the centre of the splat's
 * bounding box (A.1 step 5), not on the projection of its 3D centre -- 0.02 in alpha at the centre pixel of a strongly
 * tilted splat (oracle/surfel_autograd.py, the differentiable restatement kept as the backward oracle, follows it).
 *
 * Floating point contract of this oracle (it DEFINES the bit patterns the HIP path must reproduce for the
 * integer artefacts): IEEE-754 binary32, one rounding per written operation, evaluated left to right,
 * no FMA contraction (build with -ffp-contract=off), correctly rounded / and sqrtf.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BLOCK_X 16
#define BLOCK_Y 16
#define NEAR_N 0.2f
#define FAR_N 100.0f
#define CUTOFF 3.0f
#define FILTER_SIZE 0.707106f
#define FILTER_INV_SQUARE 2.0f

/* allmap channel offsets (upstream config.h; consumer nsr/gs_surfel.py:121-142) */
#define DEPTH_OFFSET 0
#define ALPHA_OFFSET 1
#define NORMAL_OFFSET 2
#define MIDDEPTH_OFFSET 5
#define DISTORTION_OFFSET 6

typedef struct {
    /* per-Gaussian preprocess artefacts, all sized N (or k*N) and owned by the caller */
    float *depths;          /* [N]   p_view.z                              */
    float *xy;              /* [N,2] screen-space AABB centre              */
    float *trans;           /* [N,9] Tu(3) Tv(3) Tw(3)                      */
    float *normal_opacity;  /* [N,4] view-space normal (camera facing), opacity */
    int32_t *radii;         /* [N]                                          */
    uint32_t *rect;         /* [N,4] min.x min.y max.x max.y (tile units)  */
    uint32_t *tiles_touched;/* [N]                                          */
} OraclePre;

/* number of host threads the blend loop uses (cpu_baseline reports it as "cores") */
int oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* float -> int as the CUDA/HIP hardware conversion does it (round toward zero, saturating, NaN -> 0);
 * a plain C cast is undefined out of range and x86 returns INT_MIN there. */
static inline int f2i_sat(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int)f;
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Step 1-7 of SURVEY.md A.1 (upstream preprocessCUDA + compute_transmat + compute_aabb + getRect). */
void oracle_preprocess(int N, int H, int W, const float *means3D, const float *opacities,
                       const float *scales, const float *rotations, float scale_modifier,
                       const float *viewmatrix, const float *projmatrix, OraclePre *o)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const float *vm = viewmatrix, *pm = projmatrix;
    const float halfW = (float)W / 2.0f, halfH = (float)H / 2.0f;
    const float cW = (float)(W - 1) / 2.0f, cH = (float)(H - 1) / 2.0f;
    for (int i = 0; i < N; ++i) {
        o->radii[i] = 0;
        o->tiles_touched[i] = 0;
        o->rect[4 * i + 0] = o->rect[4 * i + 1] = o->rect[4 * i + 2] = o->rect[4 * i + 3] = 0;
        const float px = means3D[3 * i], py = means3D[3 * i + 1], pz = means3D[3 * i + 2];
        /* in_frustum: p_view = [p,1] @ V ; near cull */
        const float vx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        const float vy = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        const float vz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        if (vz <= 0.2f) continue;
        /* quat (r,x,y,z) -> rotation; columns tu, tv, n.  SURVEY.md A.1 step 2: the quaternion is re-normalised by
         * 1/sqrt(|q|^2) first (upstream auxiliary.h quat_to_rotmat; its glm vec4 holds (r,x,y,z) in (.x,.y,.z,.w) and
         * sums .w^2 + .x^2 + .y^2 + .z^2 left to right).  IEEE 1.0f / sqrtf here and in the kernel. */
        const float q0 = rotations[4 * i], q1 = rotations[4 * i + 1], q2 = rotations[4 * i + 2], q3 = rotations[4 * i + 3];
        const float qs = 1.0f / sqrtf(((q3 * q3 + q0 * q0) + q1 * q1) + q2 * q2);
        const float r = q0 * qs, x = q1 * qs, y = q2 * qs, z = q3 * qs;
        const float tu[3] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y + r * z), 2.f * (x * z - r * y)};
        const float tv[3] = {2.f * (x * y - r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + r * x)};
        const float nn[3] = {2.f * (x * z + r * y), 2.f * (y * z - r * x), 1.f - 2.f * (x * x + y * y)};
        const float su = scale_modifier * scales[2 * i], sv = scale_modifier * scales[2 * i + 1];
        /* Hm rows: [su*tu,0] [sv*tv,0] [p,1];  A = Hm @ P (3x4);  M = A @ Npix (3x3) */
        float Hm[3][3] = {{tu[0] * su, tu[1] * su, tu[2] * su}, {tv[0] * sv, tv[1] * sv, tv[2] * sv}, {px, py, pz}};
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
        const float Tu[3] = {M[0][0], M[1][0], M[2][0]};
        const float Tv[3] = {M[0][1], M[1][1], M[2][1]};
        const float Tw[3] = {M[0][2], M[1][2], M[2][2]};
        /* view-space normal, made camera facing */
        float nvx = vm[0] * nn[0] + vm[4] * nn[1] + vm[8] * nn[2];
        float nvy = vm[1] * nn[0] + vm[5] * nn[1] + vm[9] * nn[2];
        float nvz = vm[2] * nn[0] + vm[6] * nn[1] + vm[10] * nn[2];
        /* transMats are written before the remaining culls upstream as well */
        for (int k = 0; k < 3; ++k) { o->trans[9 * i + k] = Tu[k]; o->trans[9 * i + 3 + k] = Tv[k]; o->trans[9 * i + 6 + k] = Tw[k]; }
        const float cs = -((vx * nvx + vy * nvy) + vz * nvz);
        if (cs == 0.0f) continue;
        const float mult = cs > 0.0f ? 1.0f : -1.0f;
        nvx = mult * nvx; nvy = mult * nvy; nvz = mult * nvz;
        /* compute_aabb, cutoff = 3 */
        const float t0 = CUTOFF * CUTOFF, t1 = CUTOFF * CUTOFF, t2 = -1.0f;
        const float d = (t0 * (Tw[0] * Tw[0]) + t1 * (Tw[1] * Tw[1])) + t2 * (Tw[2] * Tw[2]);
        if (d == 0.0f) continue;
        const float inv = 1.0f / d;
        const float f0 = inv * t0, f1 = inv * t1, f2 = inv * t2;
        const float cx = (f0 * (Tu[0] * Tw[0]) + f1 * (Tu[1] * Tw[1])) + f2 * (Tu[2] * Tw[2]);
        const float cy = (f0 * (Tv[0] * Tw[0]) + f1 * (Tv[1] * Tw[1])) + f2 * (Tv[2] * Tw[2]);
        const float hx0 = cx * cx - ((f0 * (Tu[0] * Tu[0]) + f1 * (Tu[1] * Tu[1])) + f2 * (Tu[2] * Tu[2]));
        const float hy0 = cy * cy - ((f0 * (Tv[0] * Tv[0]) + f1 * (Tv[1] * Tv[1])) + f2 * (Tv[2] * Tv[2]));
        const float ex = sqrtf(fmaxf(1e-4f, hx0)), ey = sqrtf(fmaxf(1e-4f, hy0));
        const float radius = ceilf(fmaxf(fmaxf(ex, ey), CUTOFF * FILTER_SIZE));
        /* getRect: C float->int truncation, clamped to the grid */
        const int rminx = imin(gx, imax(0, f2i_sat(((cx - radius) / BLOCK_X))));
        const int rminy = imin(gy, imax(0, f2i_sat(((cy - radius) / BLOCK_Y))));
        const int rmaxx = imin(gx, imax(0, f2i_sat(((cx + radius + BLOCK_X - 1) / BLOCK_X))));
        const int rmaxy = imin(gy, imax(0, f2i_sat(((cy + radius + BLOCK_Y - 1) / BLOCK_Y))));
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;
        o->depths[i] = vz;
        o->radii[i] = f2i_sat(radius);
        o->xy[2 * i] = cx; o->xy[2 * i + 1] = cy;
        o->normal_opacity[4 * i] = nvx; o->normal_opacity[4 * i + 1] = nvy; o->normal_opacity[4 * i + 2] = nvz;
        o->normal_opacity[4 * i + 3] = opacities[i];
        o->rect[4 * i] = rminx; o->rect[4 * i + 1] = rminy; o->rect[4 * i + 2] = rmaxx; o->rect[4 * i + 3] = rmaxy;
        o->tiles_touched[i] = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx));
    }
}

/* Binning: inclusive scan, duplicateWithKeys (y outer, x inner), stable LSD radix sort on the significant
 * key bits, identifyTileRanges.  keys/vals/tmp arrays have capacity >= D (query with oracle_count). */
int64_t oracle_count(int N, const uint32_t *tiles_touched)
{
    int64_t D = 0;
    for (int i = 0; i < N; ++i) D += tiles_touched[i];
    return D;
}

void oracle_bin(int N, int H, int W, const OraclePre *o, int64_t D, uint64_t *keys, uint32_t *vals,
                uint64_t *keys_tmp, uint32_t *vals_tmp, uint32_t *ranges /* [tiles,2] */)
{
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    const int tiles = gx * gy;
    int64_t off = 0;
    for (int i = 0; i < N; ++i) {
        if (o->radii[i] <= 0) continue;
        const uint32_t *rc = o->rect + 4 * i;
        for (uint32_t y = rc[1]; y < rc[3]; ++y)
            for (uint32_t x = rc[0]; x < rc[2]; ++x) {
                uint64_t key = (uint64_t)(y * (uint32_t)gx + x);
                key <<= 32;
                key |= (uint64_t)f2u(o->depths[i]);
                keys[off] = key; vals[off] = (uint32_t)i; ++off;
            }
    }
    /* stable LSD radix sort, 8-bit digits, over bits [0, 32 + msb(tiles)) */
    int bit = 0; { uint32_t n = (uint32_t)tiles; while (n) { ++bit; n >>= 1; } }
    const int end_bit = 32 + bit;
    uint64_t *ka = keys, *kb = keys_tmp; uint32_t *va = vals, *vb = vals_tmp;
    for (int shift = 0; shift < end_bit; shift += 8) {
        int64_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        for (int64_t j = 0; j < D; ++j) cnt[((ka[j] >> shift) & 0xFF) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t j = 0; j < D; ++j) { int64_t p = cnt[(ka[j] >> shift) & 0xFF]++; kb[p] = ka[j]; vb[p] = va[j]; }
        uint64_t *kt = ka; ka = kb; kb = kt; uint32_t *vt = va; va = vb; vb = vt;
    }
    if (ka != keys) { memcpy(keys, ka, (size_t)D * 8); memcpy(vals, va, (size_t)D * 4); }
    memset(ranges, 0, (size_t)tiles * 2 * sizeof(uint32_t));
    for (int64_t j = 0; j < D; ++j) {
        const uint32_t cur = (uint32_t)(keys[j] >> 32);
        if (j == 0) ranges[2 * cur] = 0;
        else { const uint32_t prev = (uint32_t)(keys[j - 1] >> 32); if (cur != prev) { ranges[2 * prev + 1] = (uint32_t)j; ranges[2 * cur] = (uint32_t)j; } }
        if (j == D - 1) ranges[2 * cur + 1] = (uint32_t)D;
    }
}

/* Arithmetic type of the blend.  float = the oracle proper.  -DORACLE_BLEND_F64 builds the SAME loop with every per-pair
 * operation in double (inputs are still the fp32 preprocess artefacts; outputs rounded to float at the end): a yardstick for how far
 * ANY fp32 evaluation order -- this file's included -- sits from exact arithmetic at a pixel (threshold decisions that flip,
 * cancellation in the cross product of nearly edge-on splats).  tools/parity_maxabs.py prints HIP-vs-this beside this-vs-f64. */
#ifdef ORACLE_BLEND_F64
typedef double breal;
#define BEXP exp
#define BMIN fmin
#else
typedef float breal;
#define BEXP expf
#define BMIN fminf
#endif

/* Per-tile front-to-back blend (upstream renderCUDA, RENDER_AXUTILITY=1, DUAL_VISIABLE=1).
 * pair_count (optional): number of (pixel, list entry) evaluations actually started, i.e. the real K. */
void oracle_blend(int N, int H, int W, const OraclePre *o, const float *colors, const float *bg,
                  const uint32_t *point_list, const uint32_t *ranges, float *out_color, float *out_others,
                  float *final_T, uint32_t *n_contrib, uint32_t *n_walked, int64_t *pair_count)
{
    (void)N;
    const int gx = (W + BLOCK_X - 1) / BLOCK_X, gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    int64_t pairs = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : pairs)
    for (int tile = 0; tile < gx * gy; ++tile) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ++ly)
            for (int lx = 0; lx < BLOCK_X; ++lx) {
                const int pxi = tx * BLOCK_X + lx, pyi = ty * BLOCK_Y + ly;
                if (pxi >= W || pyi >= H) continue;
                const breal pxf = (breal)pxi, pyf = (breal)pyi;
                breal T = 1.0f, C[3] = {0, 0, 0}, Nr[3] = {0, 0, 0};
                breal Dp = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
                uint32_t contributor = 0, last_contributor = 0;
                for (uint32_t j = r0; j < r1; ++j) {
                    const uint32_t id = point_list[j];
                    ++contributor; ++pairs;
                    const float *Tu = o->trans + 9 * id, *Tv = Tu + 3, *Tw = Tu + 6;
                    const breal kx = pxf * Tw[0] - Tu[0], ky = pxf * Tw[1] - Tu[1], kz = pxf * Tw[2] - Tu[2];
                    const breal lx_ = pyf * Tw[0] - Tv[0], ly_ = pyf * Tw[1] - Tv[1], lz_ = pyf * Tw[2] - Tv[2];
                    const breal p0 = ky * lz_ - kz * ly_, p1 = kz * lx_ - kx * lz_, p2 = kx * ly_ - ky * lx_;
                    if (p2 == 0.0f) continue;
                    const breal sx = p0 / p2, sy = p1 / p2;
                    const breal rho3d = sx * sx + sy * sy;
                    const breal dx = o->xy[2 * id] - pxf, dy = o->xy[2 * id + 1] - pyf;
                    const breal rho2d = FILTER_INV_SQUARE * (dx * dx + dy * dy);
                    const breal rho = BMIN(rho3d, rho2d);
                    const breal depth = (rho3d <= rho2d) ? (sx * Tw[0] + sy * Tw[1]) + Tw[2] : Tw[2];
                    if (depth < NEAR_N) continue;
                    const float *no = o->normal_opacity + 4 * id;
                    const breal power = -0.5f * rho;
                    if (power > 0.0f) continue;
                    const breal alpha = BMIN((breal)0.99f, no[3] * BEXP(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const breal test_T = T * (1 - alpha);
                    if (test_T < 0.0001f) break; /* done */
                    const breal w = alpha * T;
                    const breal A = 1 - T;
                    const breal m = FAR_N / (FAR_N - NEAR_N) * (1 - NEAR_N / depth);
                    distortion += (m * m * A + M2 - 2 * m * M1) * w;
                    Dp += depth * w;
                    M1 += m * w;
                    M2 += m * m * w;
                    if (T > 0.5f) median_depth = depth;
                    for (int ch = 0; ch < 3; ++ch) Nr[ch] += no[ch] * w;
                    for (int ch = 0; ch < 3; ++ch) C[ch] += colors[3 * id + ch] * w;
                    T = test_T;
                    last_contributor = contributor;
                }
                const size_t pid = (size_t)pyi * W + pxi, HW = (size_t)H * W;
                if (final_T) final_T[pid] = (float)T;
                if (n_contrib) n_contrib[pid] = last_contributor;
                if (n_walked) n_walked[pid] = contributor; /* list entries visited, the stopping one included */
                for (int ch = 0; ch < 3; ++ch) out_color[ch * HW + pid] = (float)(C[ch] + T * bg[ch]);
                out_others[DEPTH_OFFSET * HW + pid] = (float)Dp;
                out_others[ALPHA_OFFSET * HW + pid] = (float)(1 - T);
                for (int ch = 0; ch < 3; ++ch) out_others[(NORMAL_OFFSET + ch) * HW + pid] = (float)Nr[ch];
                out_others[MIDDEPTH_OFFSET * HW + pid] = (float)median_depth;
                out_others[DISTORTION_OFFSET * HW + pid] = (float)distortion;
            }
    }
    if (pair_count) *pair_count = pairs;
}
