#!/usr/bin/env python3
"""Exports the small fixtures under tests/golden/.  Run ONCE in the build container (needs /root/reference):

    python tests/golden/make_golden.py

What it writes and where the numbers come from:

* cameras_eval8.npz -- ``eval_pose.pt[:8]`` pushed through the REFERENCE's own camera code, imported from
  /root/reference (``utils/gs_utils/graphics_utils.py`` getWorld2View2 / getProjectionMatrix / focal2fov, composed
  exactly as ``FlowMatchingEngine_gs.c_to_3dgs_format`` does, nsr/lsgm/flow_matching_trainer.py:2174-2228).  These
  matrices PIN gaussiananything_amd.cameras (tests/test_cameras.py) against the reference.
* fps_clouds.npz -- the eight in-tree ``fps-4096.ply`` point clouds (data, float32 [8,4096,3]); the config-#2 scene
  recipe of SURVEY.md section 8d is built on them.
* surfel_cfg1_oracle.npz -- FROZEN ORACLE OUTPUT (not reference output: the reference rasterizer is a CUDA-only
  third-party extension, see oracle/surfel_raster.c "PARITY UNPINNED") for config #1 (1 000 random surfels, 256x256,
  camera 0): integer artefacts in full and the images as float16 + float64 checksums.  Guards the oracle itself
  against drift.
"""
import importlib.util
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _ref_graphics_utils():
    spec = importlib.util.spec_from_file_location("ref_graphics_utils", os.path.join(REF, "utils/gs_utils/graphics_utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def export_cameras():
    gu = _ref_graphics_utils()
    poses = torch.load(os.path.join(REF, "eval_pose.pt"))[:8].numpy()
    cv, cvp, cp = [], [], []
    tanfov = None
    for pose in poses:
        # composition of flow_matching_trainer.py:2174-2228 on top of the reference's graphics_utils functions
        c2w = pose[:16].reshape(4, 4)
        w2c = np.linalg.inv(c2w)
        R = np.transpose(w2c[:3, :3]); T = w2c[:3, 3]
        fov = gu.focal2fov(pose[16], 1)
        tanfov = math.tan(fov * 0.5)
        wvt = torch.tensor(gu.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fov, fovY=fov).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        cv.append(wvt.numpy()); cvp.append(full.numpy()); cp.append(wvt.inverse()[3, :3].numpy())
    np.savez(os.path.join(HERE, "cameras_eval8.npz"), poses=poses.astype(np.float32), cam_view=np.stack(cv),
             cam_view_proj=np.stack(cvp), cam_pos=np.stack(cp), tanfov=np.float64(tanfov))


def read_ply_xyz(path):
    with open(path, "rb") as f:
        header = b""
        while not header.endswith(b"end_header\n"):
            header += f.readline()
        lines = header.decode().split("\n")
        assert any("binary_little_endian" in l for l in lines)
        n = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
        props = [l.split()[1:] for l in lines if l.startswith("property")]
        assert [p[1] for p in props[:3]] == ["x", "y", "z"], props
        dt = {"float": "<f4", "double": "<f8", "uchar": "u1"}
        rec = np.dtype([(p[1], dt[p[0]]) for p in props])
        data = np.frombuffer(f.read(n * rec.itemsize), dtype=rec, count=n)
    return np.stack([data["x"], data["y"], data["z"]], 1).astype(np.float32)


def export_clouds():
    base = os.path.join(REF, "assets/demo-image-for-i23d/FPS_PCD_for_vae/Animals/0")
    xyz = np.stack([read_ply_xyz(os.path.join(base, d, "fps-4096.ply")) for d in sorted(os.listdir(base))])
    assert xyz.shape == (8, 4096, 3), xyz.shape
    np.savez_compressed(os.path.join(HERE, "fps_clouds.npz"), xyz=xyz)


def export_oracle_cfg1():
    from gaussiananything_amd import synthetic
    from oracle import surfel as osurf
    cams = synthetic.eval_cameras(1)
    g = synthetic.random_surfels(1000, seed=0)[0]
    m, o, s, r, c = synthetic.split_gaussians(g)
    out = osurf.rasterize(m.numpy(), o.numpy(), c.numpy(), s.numpy(), r.numpy(), cams["cam_view"][0].numpy(),
                          cams["cam_view_proj"][0].numpy(), np.ones(3, np.float32), 256, 256)
    np.savez_compressed(os.path.join(HERE, "surfel_cfg1_oracle.npz"), radii=out["radii"], rect=out["rect"],
                        tiles_touched=out["tiles_touched"], point_list=out["point_list"], ranges=out["ranges"],
                        n_contrib=out["n_contrib"].astype(np.uint16), color_f16=out["color"].astype(np.float16),
                        allmap_f16=out["allmap"].astype(np.float16),
                        color_sum=out["color"].astype(np.float64).sum((1, 2)),
                        allmap_sum=out["allmap"].astype(np.float64).sum((1, 2)), D=out["D"], pairs=out["pairs"])


if __name__ == "__main__":
    export_cameras()
    export_clouds()
    export_oracle_cfg1()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
