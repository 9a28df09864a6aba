// surfel_api.hip -- C-ABI entry points of the surfel rasterizer (declared in include/ga_surfel.h).
// Host-side only: argument validation, workspace carving, and the launch sequence on the caller's stream.
#include <stdlib.h>

#include "surfel_common.h"

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool make_dims(int32_t N, int32_t V, int32_t H, int32_t W, ga::Dims *d)
{
    if (N < 0 || V < 1 || H < 1 || W < 1) return false;
    d->N = N; d->V = V; d->H = H; d->W = W;
    d->gx = (W + ga::kTile - 1) / ga::kTile;
    d->gy = (H + ga::kTile - 1) / ga::kTile;
    if (d->gx > 65535 || d->gy > 65535) return false;
    const int64_t tiles = (int64_t)d->gx * d->gy;
    if (tiles * V >= (1 << 24)) return false;
    if ((int64_t)N * V >= ((int64_t)1 << 31)) return false;
    d->tiles = (int)tiles;
    return true;
}

}  // namespace

extern "C" {

const char *ga_surfel_version(void) { return "ga_mi355 surfel gfx950 r6"; }

int ga_surfel_workspace_layout(int32_t num_points, int32_t num_views, int32_t image_height, int32_t image_width,
                               int64_t capacity, GaSurfelWorkspaceLayout *out)
{
    return ga_surfel_workspace_layout2(num_points, num_views, image_height, image_width, capacity, 0, out);
}

int ga_surfel_workspace_layout2(int32_t num_points, int32_t num_views, int32_t image_height, int32_t image_width,
                                int64_t capacity, int64_t seg_capacity, GaSurfelWorkspaceLayout *out)
{
    ga::Dims d;
    if (!out) return GA_ERR_NULL_ARG;
    if (!make_dims(num_points, num_views, image_height, image_width, &d) || capacity < 0 ||
        capacity > 0xFFFFFFFFll || seg_capacity < 0 || seg_capacity > 0xFFFFFFll)
        return GA_ERR_BAD_SHAPE;
    const size_t nv = (size_t)d.N * d.V, nt = (size_t)d.V * d.tiles, cap = (size_t)capacity;
    size_t off = 0;
    out->status = off;      off += align256(GA_STATUS_WORDS * sizeof(int64_t));
    out->seg_sync = off;    off += align256(8 * (cap / 1024 + 1) * 4);
    out->tile_count = off;  off += align256(nt * 4);
    out->view_total = off;  off += align256((size_t)d.V * ga::kViewSlots * 8);
    out->tile_cursor = off; off += align256(nt * 4);
    out->tile_start = off;  off += align256((nt + 1) * 4);   /* everything in front of it is cleared by the memset of a forward */
    out->tile_order = off;  off += align256(nt * 16);   /* uint4 (tile, begin, length, 0) per schedule slot */
    out->run_table = off;   off += align256((cap / GA_SURFEL_SORT_RUN + 1) * 16);  /* uint4 (tile, run, begin, length) */
    out->rect = off;        off += align256(nv * 4 * sizeof(uint16_t));
    out->depth = off;       off += align256(nv * 4);
    out->record = off;      off += align256(nv * ga::kRec * 4);
    out->keys = off;        off += align256(cap * 8);
    out->point_list = off;  off += align256(cap * 4);
    out->seg_table = off;   off += align256(ga::kSegTableWords * 4);   /* 2 x 40 class entries + the launch epoch word */
    out->seg_scratch = off; off += align256((size_t)ga::seg_items(capacity, seg_capacity) * (size_t)ga::kSegFloats * 8);
    out->total_bytes = off;
    return GA_OK;
}

int ga_surfel_forward(const GaSurfelForwardArgs *a, void *stream_v)
{
    if (!a) return GA_ERR_NULL_ARG;
    ga::Dims d;
    if (!make_dims(a->num_points, a->num_views, a->image_height, a->image_width, &d)) return GA_ERR_BAD_SHAPE;
    GaSurfelWorkspaceLayout L;
    const int rc = ga_surfel_workspace_layout2(d.N, d.V, d.H, d.W, a->capacity, a->seg_capacity, &L);
    if (rc != GA_OK) return rc;
    if (!a->viewmatrix || !a->projmatrix || !a->bg || !a->out_color || !a->out_others || !a->workspace)
        return GA_ERR_NULL_ARG;
    if (d.N > 0 && (!a->means3D || !a->opacities || !a->colors || !a->scales || !a->rotations || !a->radii))
        return GA_ERR_NULL_ARG;
    if (a->workspace_bytes < L.total_bytes) return GA_ERR_WORKSPACE;
    if (((uintptr_t)a->workspace & 255) != 0) return GA_ERR_WORKSPACE;
    if (a->seg_T && a->seg_T_floats < (a->capacity / 128 + (int64_t)d.V * d.tiles + 1) * 256) return GA_ERR_WORKSPACE;

    hipStream_t s = reinterpret_cast<hipStream_t>(stream_v);
    unsigned char *w = static_cast<unsigned char *>(a->workspace);
    ga::Workspace ws;
    ws.status = reinterpret_cast<int64_t *>(w + L.status);
    ws.seg_sync = reinterpret_cast<uint32_t *>(w + L.seg_sync);
    ws.seg_table = reinterpret_cast<uint32_t *>(w + L.seg_table);
    ws.seg_scratch = reinterpret_cast<unsigned long long *>(w + L.seg_scratch);
    ws.tile_count = reinterpret_cast<uint32_t *>(w + L.tile_count);
    ws.tile_start = reinterpret_cast<uint32_t *>(w + L.tile_start);
    ws.tile_cursor = reinterpret_cast<uint32_t *>(w + L.tile_cursor);
    ws.view_total = reinterpret_cast<unsigned long long *>(w + L.view_total);
    ws.tile_order = reinterpret_cast<uint4 *>(w + L.tile_order);
    ws.run_table = reinterpret_cast<uint4 *>(w + L.run_table);
    ws.rect = reinterpret_cast<uint16_t *>(w + L.rect);
    ws.depth = reinterpret_cast<float *>(w + L.depth);
    ws.record = reinterpret_cast<float *>(w + L.record);
    ws.keys = reinterpret_cast<uint64_t *>(w + L.keys);
    ws.point_list = reinterpret_cast<uint32_t *>(w + L.point_list);

    (void)hipGetLastError();
    // (event 0 is recorded after this memset so that it brackets kernels only)
    // status words, segment flags / counters and tile counters are contiguous at the head of the workspace: one memset
    // node clears them all
    // -- unless the caller vouches that the previous forward on this workspace left them clean (the tile scan clears what it and
    // the blend accumulate into at its end)
    if (!(a->flags & GA_SURFEL_FLAG_WORKSPACE_CLEAN) && hipMemsetAsync(w + L.status, 0, L.tile_start - L.status, s) != hipSuccess)
        return GA_ERR_LAUNCH;
    auto mark = [&](int k) {
        if (a->stage_events && a->stage_events[k]) (void)hipEventRecord(static_cast<hipEvent_t>(a->stage_events[k]), s);
    };
    mark(0);
    if (d.N > 0) ga::launch_preprocess(*a, d, ws, s);
    mark(1);
    ga::launch_binning(*a, d, ws, s);
    mark(2);
    ga::launch_tile_sort(*a, d, ws, s);
    mark(3);
    ga::launch_blend(*a, d, ws, s);
    mark(4);
    return hipGetLastError() == hipSuccess ? GA_OK : GA_ERR_LAUNCH;
}

}  // extern "C"
