/*
 * ga_surfel.h -- C-ABI of the MI355X-native 2D-surfel ("2DGS") rasterizer forward.
 *
 * Drop-in boundary for the render half of GaussianAnything's render-and-denoise hot path.  A binding of this
 * ABI replaces the third-party CUDA extension `diff_surfel_rasterization._C.rasterize_gaussians` that the
 * reference reaches through
 *     GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
 *                                         cov3D_precomp)            /root/reference/nsr/gs_surfel.py:85-114
 * and, one level up, the per-(batch, view) Python loop of
 *     GaussianRenderer2DGS.render(...)                              /root/reference/nsr/gs_surfel.py:41-202
 * (one call here rasterizes ALL views of one Gaussian set; see INTEGRATION.md for the ctypes binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, gfx950) unless it says "host"; all tensors are contiguous fp32 /
 *     int32 exactly as the reference passes them; the caller owns every buffer, nothing is allocated inside;
 *   - all work is enqueued on `stream`; there is no host synchronisation, no internal thread, no exception: the
 *     return value is 0 or a negative GA_ERR_* code for host-detectable argument errors; run-time conditions that
 *     only the device can see (binned-list overflow) are reported through the device-side `status` words;
 *   - matrices are the reference's row-vector 4x4 (`p_view = [p,1] @ viewmatrix`, 16 floats row-major).
 */
#ifndef GA_SURFEL_H
#define GA_SURFEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GA_OK 0
#define GA_ERR_NULL_ARG (-1)       /* a required pointer is NULL                                   */
#define GA_ERR_BAD_SHAPE (-2)      /* N, V, H, W or capacity out of range (H, W <= 4096*16, V*tiles < 2^24) */
#define GA_ERR_WORKSPACE (-3)      /* workspace_bytes smaller than ga_surfel_workspace_layout() says */
#define GA_ERR_LAUNCH (-4)         /* a HIP launch failed (hipGetLastError is left set)            */

/* status words written by the device (int64 each) */
#define GA_STATUS_NUM_RENDERED 0   /* D = sum over views of tiles touched (upstream `num_rendered`) */
#define GA_STATUS_OVERFLOW 1       /* 1 if D > capacity or SEG_WORK > seg_capacity: outputs are then NOT written; retry with more */
#define GA_STATUS_MAX_TILE 2       /* longest per-tile list (diagnostic)                           */
#define GA_STATUS_EXTRA_RUNS 3     /* number of entries of run_table (internal)                    */
#define GA_STATUS_BLEND_ITERS 4   /* with GA_SURFEL_FLAG_STATS: total inner-loop iterations of the blend (all waves) */
#define GA_STATUS_BLEND_MAX_ITERS 5 /* with GA_SURFEL_FLAG_STATS: most iterations executed by one wave              */
#define GA_STATUS_BLEND_CHUNKS 6  /* with GA_SURFEL_FLAG_STATS: total 64-entry chunks consumed                          */
#define GA_STATUS_LONG_TILES 7    /* tiles whose list is blended in segments (internal): they lead tile_order          */
#define GA_STATUS_BLEND_LANE_SLOTS 8 /* with GA_SURFEL_FLAG_STATS: (pixel, pair) evaluations that had work (of 64 x BLEND_ITERS) */
#define GA_STATUS_SEG_WORK 9      /* (tile, segment) work items of the segmented tiles (internal)                      */
#define GA_STATUS_SEG_TICKET 10   /* next segment work item to hand out (internal)                                     */
#define GA_STATUS_BLEND_LANE_MAX 11 /* with GA_SURFEL_FLAG_STATS: sum over (wave, work item, pass) of the LONGEST per-pixel survivor
                                     list of the wave: what BLEND_ITERS would be if the 64 pixels did not wait for each other */
#define GA_STATUS_SPLIT_A_ITERS 12 /* with GA_SURFEL_FLAG_STATS, split walk: wave-level pair-evaluation instructions (64 pair slots each) */
#define GA_STATUS_SPLIT_A_PAIRS 13 /* ... (entry, pixel) pairs evaluated                                                             */
#define GA_STATUS_SPLIT_B_ROWS 14  /* ... wave-level composite rows (64 pixel slots each)                                            */
#define GA_STATUS_SPLIT_B_ITEMS 15 /* ... items composited                                                                           */
#define GA_STATUS_WORDS 16

typedef struct GaSurfelForwardArgs {
    int32_t num_points;      /* N Gaussians                                                       */
    int32_t num_views;       /* V views rendered from the same Gaussians                          */
    int32_t image_height;    /* H                                                                 */
    int32_t image_width;     /* W                                                                 */
    float scale_modifier;    /* GaussianRasterizationSettings.scale_modifier                      */
    int32_t flags;           /* GA_SURFEL_FLAG_*                                                  */
    const float *means3D;    /* [N,3]                                                             */
    const float *opacities;  /* [N] (reference passes [N,1])                                      */
    const float *colors;     /* [N,3] colors_precomp (sh_degree 0 path; shs unsupported as in the reference call) */
    const float *scales;     /* [N,2]                                                             */
    const float *rotations;  /* [N,4] quaternion (w,x,y,z); re-normalised on the device (1 / sqrt) */
    const float *viewmatrix; /* [V,16]                                                            */
    const float *projmatrix; /* [V,16] full view-projection                                       */
    const float *bg;         /* [3]                                                               */
    float *out_color;        /* [V,3,H,W]                                                         */
    float *out_others;       /* [V,7,H,W] allmap: depth, alpha, normal xyz, median depth, distortion */
    int32_t *radii;          /* [V,N]                                                             */
    void *workspace;         /* see ga_surfel_workspace_layout                                    */
    size_t workspace_bytes;
    int64_t capacity;        /* max binned entries D the workspace was sized for                  */
    void **stage_events;     /* host, optional (NULL = none): GA_SURFEL_STAGE_EVENTS hipEvent_t handles recorded on
                                `stream` at the stage boundaries: [0] start, [1] after preprocess, [2] after tile scan +
                                fill, [3] after per-tile sort, [4] after blend.  Measurement only.          */
    int64_t seg_capacity;    /* (tile, segment) work items the exchange scratch of the segmented blend holds (lists of 2048
                                entries or more are blended in 256..512-entry segments by several workgroups); 0 = default
                                capacity / 2048 + 128.  The worst case is capacity / 256.  More work items than this is
                                reported like D > capacity: GA_STATUS_OVERFLOW, nothing rendered, the number needed in
                                GA_STATUS_SEG_WORK.  Same value as passed to ga_surfel_workspace_layout2.       */
    float *seg_T;            /* optional (NULL = not wanted; inference never sets it), for ga_surfel_backward: the blend records,
                                for every 128-entry segment k of every non-empty tile list and every pixel of the tile, the
                                transmittance the pixel enters the segment with, or -1 when its walk ended earlier (pixel outside
                                the image, stop rule).  Row floor(list begin / 128) + (view * tiles + tile) + k of 256 floats;
                                pixel (x, y) of the tile at ((y >> 3) * 2 + (x >> 3)) * 64 + (y & 7) * 8 + (x & 7).  With it the
                                backward does not walk the lists a first time for these products.  Costs the blend the
                                run-ahead across every second 64-entry chunk boundary and 4 bytes per (segment, pixel).       */
    int64_t seg_T_floats;    /* floats `seg_T` holds: at least (capacity / 128 + V * tiles + 1) * 256 (else GA_ERR_WORKSPACE) */
} GaSurfelForwardArgs;

#define GA_SURFEL_STAGE_EVENTS 5

#define GA_SURFEL_FLAG_NONE 0
#define GA_SURFEL_FLAG_STATS 1      /* collect the GA_STATUS_BLEND_* diagnostics (costs three atomics per wave) */
#define GA_SURFEL_FLAG_WORKSPACE_CLEAN 2 /* the LAST use of this workspace was a ga_surfel_forward with the same sizes that has been
                                       enqueued completely (it returned GA_OK): its tile scan left the accumulating words of the
                                       workspace head zeroed, so the clearing memset in front of this forward is skipped.  Never
                                       set it for the first forward on a workspace (or after writing to it) */

#define GA_SURFEL_FLAG_SPLIT_WALK 4  /* blend unsegmented lists with the SPLIT walk (evaluate with lanes = (entry, pixel) pairs, composite
                                       with lanes = pixels; surfel_blend.hip) instead of the fused lanes = pixels walk.  Built, parity
                                       green and measured in round 4 -- 175 us against 140 us on BASELINE configs[1] -- hence opt-in */
#define GA_SURFEL_FLAG_BG_IN_BLEND 8  /* round 6, A/B aid: the background pixels of the EMPTY tiles (two thirds of the tiles at BASELINE configs[1]:
                                         56 MB of stores) are written by the blend launch's own workgroups as in rounds 1-5; default: by the
                                         otherwise idle waves of the per-tile sort launch in front of it, whose lists those tiles do not have */

/* Byte offsets of the workspace sections (all 256-byte aligned).  Tests read the integer artefacts
 * (rect, tile ranges, sorted point list) straight out of the workspace through these offsets. */
typedef struct GaSurfelWorkspaceLayout {
    size_t status;      /* int64[GA_STATUS_WORDS]                                                  */
    size_t seg_sync;    /* uint32[8*(capacity/1024+1)] per segmented tile: four per-quadrant arrival counters and the
                           saturation word of the segmented blend; cleared with the status words */
    size_t tile_count;  /* uint32[V*tiles]   entries per (view, tile)                             */
    size_t tile_start;  /* uint32[V*tiles+1] exclusive scan of tile_count                         */
    size_t tile_cursor; /* uint32[V*tiles]   scratch of the fill pass: slots of the tile's list handed out so far (zero between launches) */
    size_t tile_order;  /* uint32[V*tiles]   (view,tile) ids, longest lists first: workgroup -> tile schedule */
    size_t run_table;   /* uint32[2*(capacity/GA_SURFEL_SORT_RUN+1)] (tile id, run index) of the 2nd.. sort runs of long lists */
    size_t rect;        /* uint16[V*N*4]     tile rect min.x min.y max.x max.y (0 when culled)    */
    size_t depth;       /* float[V*N]        view-space depth (sort key)                          */
    size_t record;      /* float[V*N*GA_SURFEL_RECORD_FLOATS] blend-ready splat records           */
    size_t keys;        /* uint64[capacity]  (depth bits << 32 | gaussian index), binned per tile */
    size_t point_list;  /* uint32[capacity]  gaussian indices, per tile in (depth, index) order   */
    size_t seg_table;   /* uint32[128]       per length class: first tile_order slot, first segment work item (2 x 40 words);
                           word 96: launch epoch of the exchange words, bumped by the tile scan (never cleared; any initial
                           value) */
    size_t seg_scratch; /* uint64[seg_capacity * 15 * 256] per segment: transmittance + 14 partial sums per pixel,
                           each word (value, launch epoch) */
    size_t view_total;  /* uint64[V][64]     entries per view (sum of its tile counters) as 64 partial counts, accumulated by the
                           preprocess; in the head region that is cleared with the status words */
    size_t total_bytes;
} GaSurfelWorkspaceLayout;

#define GA_SURFEL_RECORD_FLOATS 24
#ifndef GA_SURFEL_SORT_RUN
#define GA_SURFEL_SORT_RUN 2048    /* entries sorted per LDS pass of the per-tile sort */
#endif

/* host: fills `out` for the given problem size; returns GA_OK or GA_ERR_BAD_SHAPE */
int ga_surfel_workspace_layout(int32_t num_points, int32_t num_views, int32_t image_height, int32_t image_width,
                               int64_t capacity, GaSurfelWorkspaceLayout *out);

/* as above with an explicit seg_capacity (0 = the default of ga_surfel_workspace_layout) */
int ga_surfel_workspace_layout2(int32_t num_points, int32_t num_views, int32_t image_height, int32_t image_width,
                                int64_t capacity, int64_t seg_capacity, GaSurfelWorkspaceLayout *out);

/* host: enqueue the whole forward (preprocess, tile binning, per-tile depth sort, blend) on `stream`
 * (a hipStream_t passed as void* so that this header needs no HIP include). */
int ga_surfel_forward(const GaSurfelForwardArgs *args, void *stream);

/* host: the per-pixel post-processing of GaussianRenderer2DGS.render (/root/reference/nsr/gs_surfel.py:121-163) over the
 * outputs of ga_surfel_forward, all views in one pass:
 *   image = clamp(color, 0, 1) (:163);  rend_normal = allmap[2:5] rotated view -> world, n @ viewmatrix[:3,:3]^T (:126-128);
 *   depth = nan_to_num(allmap[5], nan = 0, posinf = 0) (:133-134, depth_ratio = 1).
 * alpha = allmap[1] and dist = allmap[6] need no computation.  Device pointers, fp32, contiguous; nothing is allocated. */
typedef struct GaSurfelPostArgs {
    int32_t num_views, image_height, image_width;
    const float *color;       /* [V, 3, H, W]  out_color of ga_surfel_forward                */
    const float *allmap;      /* [V, 7, H, W]  out_others of ga_surfel_forward               */
    const float *viewmatrix;  /* [V, 4, 4]     row-vector world_view_transform (cam_view)     */
    float *image;             /* [V, 3, H, W]                                                */
    float *rend_normal;       /* [V, 3, H, W]                                                */
    float *depth;             /* [V, 1, H, W]                                                */
} GaSurfelPostArgs;
int ga_surfel_postprocess(const GaSurfelPostArgs *args, void *stream);

/* host: BACKWARD of ga_surfel_forward (the gradient the reference's training call sites take through
 * GaussianRasterizer(...)(means3D, means2D, ...), /root/reference/nsr/gs_surfel.py:104-114; upstream backward.cu is third
 * party and absent: oracle/surfel_autograd.py is the oracle).  `fwd` is the argument block of the forward call it
 * differentiates, UNCHANGED, with the workspace as that call left it (tile ranges and point lists are read again) and its
 * outputs out_color / out_others / radii.  grad_color [V,3,H,W] and grad_others [V,7,H,W] are dL/d(out_color) and
 * dL/d(out_others); the gradients with respect to the Gaussians are summed over the views.  The selection
 * min(rho3d, rho2d), the alpha / depth / transmittance tests, the 0.99 clamp, the normal's facing sign and the median depth
 * (channel 5 of out_others) are constants of the gradient.  Nothing is allocated; `scratch` (16-byte aligned) holds
 * ga_surfel_backward_scratch_bytes(&fwd) bytes: the per-(view, Gaussian) records and gradient words, and the table and
 * per-pixel exchange arrays of the 128-entry list segments the blend backward is parallel over (sized from fwd.capacity). */
typedef struct GaSurfelBackwardArgs {
    GaSurfelForwardArgs fwd;
    const float *grad_color;   /* [V,3,H,W]                                                      */
    const float *grad_others;  /* [V,7,H,W]                                                      */
    void *scratch;
    size_t scratch_bytes;
    float *grad_means3D;       /* [N,3]  (all five are overwritten)                              */
    float *grad_opacities;     /* [N]                                                            */
    float *grad_colors;        /* [N,3]                                                          */
    float *grad_scales;        /* [N,2]                                                          */
    float *grad_rotations;     /* [N,4]  with respect to the quaternion as given (not normalised) */
    /* (fwd.seg_T, when the forward was run with it: the transmittance pass and its prefix launch are skipped) */
} GaSurfelBackwardArgs;
size_t ga_surfel_backward_scratch_bytes(const GaSurfelForwardArgs *fwd);   /* 0: bad shape */
int ga_surfel_backward(const GaSurfelBackwardArgs *args, void *stream);

/* host: library identification, e.g. "ga_mi355 surfel gfx950 r2" */
const char *ga_surfel_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GA_SURFEL_H */
