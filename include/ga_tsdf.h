/*
 * ga_tsdf.h -- C-ABI of the MI355X-native TSDF fusion + mesh extraction (the last hand-off format).
 *
 * Replaces, for the mesh export of a generated object, the Open3D calls of
 *     FlowMatchingEngine_gs.extract_mesh_bounded          /root/reference/nsr/lsgm/flow_matching_trainer.py:1318-1395
 *         o3d.pipelines.integration.ScalableTSDFVolume(voxel_length, sdf_trunc, color_type=RGB8)            :1344-1348
 *         o3d.geometry.RGBDImage.create_from_color_and_depth(color u8, depth f32, depth_trunc, depth_scale=1) :1382-1388
 *         volume.integrate(rgbd, intrinsic, extrinsic)                                                      :1390
 *         volume.extract_triangle_mesh()                                                                    :1392
 * (Open3D is a third-party dependency, `open3d` unpinned in /root/reference/requirements.txt:33, absent from this image; its
 * published algorithm -- ScalableTSDFVolume / UniformTSDFVolume, Integrate and ExtractTriangleMesh -- is restated in
 * oracle/tsdf.py, parity UNPINNED, see there.)
 *
 * The volume is DENSE over a caller-chosen box of 16^3-voxel units on Open3D's unit lattice (unit (i,j,k) spans
 * [i, i+1) x [j, j+1) x [k, k+1) times 16 voxel_length; voxel centres at (n + 0.5) voxel_length): 288 GB of HBM hold the
 * whole bounding cube of an object at the reference's resolution (344^3 voxels x 20 B = 0.8 GB) many times over, so the hash
 * map of units becomes a flat array and a per-frame "touched" byte per unit.  Units outside the box are not allocated
 * (Open3D would open them); everything inside follows Open3D's arithmetic.
 *
 * Conventions as in ga_surfel.h: device pointers unless marked host, caller owns every buffer, work is enqueued on `stream`,
 * 0 or a negative GA_ERR_* code is returned, no exceptions, no host synchronisation.
 * Voxel storage is unit-blocked: voxel (x, y, z) of the box lives at ((ux Uy + uy) Uz + uz) 4096 + (z & 15) 256 + (x & 15) 16
 * + (y & 15), u* = coordinate >> 4 (integration is one thread per (x, y) column walking z, as Open3D's loop does).
 */
#ifndef GA_TSDF_H
#define GA_TSDF_H

#include <stddef.h>
#include <stdint.h>

#include "ga_surfel.h" /* GA_OK, GA_ERR_* */

#ifdef __cplusplus
extern "C" {
#endif

#define GA_TSDF_UNIT 16 /* voxels per unit edge (ScalableTSDFVolume volume_unit_resolution, Open3D's default) */

typedef struct GaTsdfVolume {
    int32_t units[3];     /* Ux, Uy, Uz units in the box (voxels = 16 U)                                         */
    int32_t unit0[3];     /* lattice index of the box's first unit (origin = unit0 * 16 * voxel_length)          */
    double voxel_length;
    double sdf_trunc;
    float *tsdf;          /* [nvox]    nvox = 4096 Ux Uy Uz; zero-initialised by the caller                      */
    float *weight;        /* [nvox]    zero-initialised (weight 0 = never observed)                              */
    float *color;         /* [3][nvox] running mean of the 8-bit colours (Open3D keeps doubles; fp32 here), zero-initialised */
    uint8_t *touched;     /* [Ux Uy Uz] scratch: units opened by the frame being integrated                      */
    uint8_t *allocated;   /* [Ux Uy Uz] units ever opened; zero-initialised                                      */
} GaTsdfVolume;

/* One RGB-D frame, as the reference prepares it (flow_matching_trainer.py:1371-1388): colour clipped to [0,1] and
 * truncated to 8 bits, depth zeroed where alpha < alpha_thres (alpha may be NULL) or depth >= depth_trunc. */
typedef struct GaTsdfFrame {
    int32_t height, width;
    const float *rgb;      /* [3,H,W] */
    const float *depth;    /* [H,W]   */
    const float *alpha;    /* [H,W] or NULL */
    float alpha_thres;
    float depth_trunc;
    double fx, fy, cx, cy;      /* pinhole intrinsics in pixels (utils/mesh_util.py:80-110 to_cam_open3d_compat)      */
    double extrinsic[16];       /* host values: world -> camera, row-major 4x4 (= cam_view^T of the reference)          */
    double pose[16];            /* host values: its inverse                                                            */
    int32_t depth_sampling_stride; /* ScalableTSDFVolume default 4: pixels sampled when opening units                   */
} GaTsdfFrame;

/* fuse one frame: open the units within sdf_trunc of the sampled depth points, integrate their voxels */
int ga_tsdf_integrate(const GaTsdfVolume *volume, const GaTsdfFrame *frame, void *stream);

/* mesh extraction (marching cubes over cubes whose 8 corners have been observed; vertices on the zero crossings of the
 * intersected edges, shared between cubes; triangles by the derived case table, tools/gen_mc_table.py).
 *   ga_tsdf_mesh_count : classifies and counts; device `counts[0]` = vertices, `counts[1]` = triangles.
 *   ga_tsdf_mesh_emit  : after ga_tsdf_mesh_count on the same scratch; writes vertices [nv,3] (world coordinates), colors
 *                        [nv,3] in [0,1], triangles [nt,3] (vertex indices); order: units in lexicographic order, voxels in
 *                        storage order, edges x, y, z.  Pass the counts read back from the device; nothing is written
 *                        past them. */
size_t ga_tsdf_mesh_scratch_bytes(const GaTsdfVolume *volume);
int ga_tsdf_mesh_count(const GaTsdfVolume *volume, void *scratch, size_t scratch_bytes, int64_t *counts, void *stream);
int ga_tsdf_mesh_emit(const GaTsdfVolume *volume, void *scratch, size_t scratch_bytes, int64_t num_vertices, int64_t num_triangles,
                      float *vertices, float *colors, int32_t *triangles, void *stream);

/* Connected triangle clusters of post_process_mesh (/root/reference/utils/mesh_util.py:22-44 -> Open3D's cluster_connected_triangles:
 * triangles joined through shared edges): pair_a[i], pair_b[i] (device, int64) are two triangles that share an edge; on return
 * labels[t] (device, int32 [num_triangles]) is the SMALLEST triangle index of t's cluster -- one lock-free union-find pass, the same
 * labels whatever order the pairs are served in. */
int ga_mesh_cluster_labels(const int64_t *pair_a, const int64_t *pair_b, int64_t num_pairs, int32_t *labels, int64_t num_triangles,
                           void *stream);

/* host: write a triangle mesh (HOST arrays: vertices [nv,3], colors [nv,3] in [0,1] or NULL, triangles [nt,3] zero-based) as
 * Wavefront OBJ with per-vertex colours, replacing o3d.io.write_triangle_mesh (flow_matching_trainer.py:1297, 1311).
 * GA_ERR_LAUNCH reports an I/O failure. */
int ga_mesh_write_obj(const char *path, const float *vertices, const float *colors, const int32_t *triangles,
                      int64_t num_vertices, int64_t num_triangles);

#ifdef __cplusplus
}
#endif
#endif
