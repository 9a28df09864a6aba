#!/usr/bin/env python3
"""Generates tests/golden/loop_ref.pt: the REFERENCE'S OWN host code of the hot path run in the loop, here in the build
container (the GPU box has no /root/reference, so the GPU tests replay these fixtures).

  render   /root/reference/nsr/gs_surfel.py:41-202  GaussianRenderer2DGS.render, imported as it stands, with stand-ins only for
           the absent imports (kiui, utils.point_utils) and a RECORDING `diff_surfel_rasterization`: every
           GaussianRasterizationSettings / rasterizer call the reference issues is recorded (what a drop-in rasterizer
           receives), the rasterizer returns seeded pseudo-renders (colours outside [0, 1], NaN and zero-alpha pixels), and the
           dict the reference builds from them is the golden output of the renderer-level post-processing.
  sample   FlowMatchingEngine.sample (/root/reference/nsr/lsgm/flow_matching_trainer.py:700-744) restated line by line around
           the reference's own transport/*.py (Sampler.sample_ode, integrators.ode; the oracle integrator stands in for the
           absent torchdiffeq) and the reference's own DiT classes with the weights of tests/golden/dit_ref_stage{1,2}.pt:
           CPU-seeded noise -> bf16 -> CFG batch -> sample_ode -> last state -> conditional half, stage 1, the x 0.164 / clip /
           / 0.45 hand-off (:987-1000, :1079; sgm PCD_Scaler), stage 2 with uc == c.  Euler trajectories are kept state by state.

Run once:  python tests/golden/make_loop_golden.py
"""
import contextlib
import importlib
import io
import os
import sys
import types
from collections import namedtuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_dit_loader as L  # noqa: E402


# ---- render ----------------------------------------------------------------------------------------------------------
def make_render():
    from gaussiananything_amd import synthetic
    calls = []
    Settings = namedtuple("GaussianRasterizationSettings", ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                            "viewmatrix", "projmatrix", "sh_degree", "campos", "prefiltered", "debug"])

    class Rasterizer:
        def __init__(self, raster_settings):
            self.s = raster_settings

        def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
            k = len(calls)
            g = torch.Generator().manual_seed(1000 + k)
            H, W = self.s.image_height, self.s.image_width
            color = torch.rand(3, H, W, generator=g) * 1.6 - 0.3                 # outside [0, 1]: the clamp matters
            allmap = torch.randn(7, H, W, generator=g)
            allmap[1] = torch.rand(H, W, generator=g)                            # alpha
            allmap[1][torch.rand(H, W, generator=g) < 0.2] = 0.0                 # empty pixels
            allmap[5][torch.rand(H, W, generator=g) < 0.15] = float("nan")       # median depth of pixels nothing reached
            allmap[5][0, 0], allmap[5][0, 1] = float("inf"), float("-inf")
            radii = torch.randint(0, 9, (means3D.shape[0],), generator=g, dtype=torch.int32)
            calls.append(dict(settings={f: getattr(self.s, f) for f in Settings._fields},
                              args=dict(means3D=means3D.clone(), means2D=means2D.clone(), opacities=opacities.clone(), shs=shs,
                                        colors_precomp=colors_precomp.clone(), scales=scales.clone(), rotations=rotations.clone(),
                                        cov3D_precomp=cov3D_precomp),
                              out=dict(color=color.clone(), radii=radii.clone(), allmap=allmap.clone())))
            return color, radii, allmap

    saved = dict(sys.modules)
    sys.modules["kiui"] = types.ModuleType("kiui")
    sys.modules["diff_surfel_rasterization"] = types.SimpleNamespace(GaussianRasterizationSettings=Settings, GaussianRasterizer=Rasterizer)
    pu = types.ModuleType("utils.point_utils")
    pu.depth_to_normal = pu.depth_to_normal_2 = lambda *a, **k: None   # imported by the module, not used by render()
    ut = types.ModuleType("utils")
    ut.point_utils = pu
    sys.modules["utils"], sys.modules["utils.point_utils"] = ut, pu
    try:
        spec = importlib.util.spec_from_file_location("ref_gs_surfel", os.path.join(REF, "nsr/gs_surfel.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k in ("kiui", "diff_surfel_rasterization", "utils", "utils.point_utils"):
            if k in saved:
                sys.modules[k] = saved[k]
            else:
                sys.modules.pop(k, None)
    r = object.__new__(mod.GaussianRenderer2DGS)     # __init__ puts bg_color on "cuda": set the attributes it would set
    r.bg_color = torch.tensor([1, 1, 1], dtype=torch.float32)
    r.output_size, r.out_chans, r.rendering_kwargs = 24, 3, {}
    B, V, N = 2, 3, 40
    cams = synthetic.eval_cameras(8)
    gs = torch.stack([synthetic.random_surfels(N, seed=70 + b)[0] for b in range(B)], 0)
    idx = torch.tensor([[0, 2, 5], [1, 4, 7]])
    cam_view, cam_view_proj, cam_pos = cams["cam_view"][idx], cams["cam_view_proj"][idx], cams["cam_pos"][idx]
    tanfov = float(cams["tanfov"])
    out_default = r.render(gs, cam_view, cam_view_proj, cam_pos, tanfov)
    n_default = len(calls)
    bg = torch.tensor([0.2, 0.4, 0.6])
    out_args = r.render(gs[:1], cam_view[:1], cam_view_proj[:1], cam_pos[:1], tanfov, bg_color=bg, scale_modifier=0.7, output_size=16)
    return dict(gaussians=gs, cam_view=cam_view, cam_view_proj=cam_view_proj, cam_pos=cam_pos, tanfov=tanfov, calls=calls,
                n_default=n_default, out_default=out_default, bg=bg, scale_modifier=0.7, output_size=16, out_args=out_args)


# ---- engine.sample over the reference's transport and DiT classes -----------------------------------------------------------
def make_sample():
    from oracle import ode as oo
    m = L.install()
    nfe = {"n": 0}

    def fake_odeint(fn, x, t, method, atol, rtol):
        stats = {}
        y = oo.odeint(lambda ts, yy: fn(torch.tensor(ts, dtype=torch.float32), torch.from_numpy(yy).float()).numpy().astype(np.float64),
                      x.numpy().astype(np.float64), t.numpy().astype(np.float64), method=method, atol=float(atol[0]), rtol=float(rtol[0]),
                      stats=stats)
        nfe["n"] = stats.get("nfe", 0)
        return torch.from_numpy(y).float()

    saved = {k: sys.modules.get(k) for k in ("torchdiffeq", "sgm", "sgm.util", "transport")}
    sys.modules["torchdiffeq"] = types.SimpleNamespace(odeint=fake_odeint)
    if "sgm" not in sys.modules:
        sys.modules["sgm"] = types.ModuleType("sgm")
        sys.modules["sgm.util"] = types.SimpleNamespace(instantiate_from_config=lambda *a, **k: None)
    for k in [k for k in sys.modules if k == "transport" or k.startswith("transport.")]:
        del sys.modules[k]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import transport as ref_transport
    rt = ref_transport.create_transport("GVP", "velocity", None, None, None, snr_type="uniform")
    try:
        sampler = ref_transport.Sampler(rt, guider_config=None)
    except TypeError:
        sampler = ref_transport.Sampler(rt)

    def load(stage):
        d = torch.load(os.path.join(HERE, f"dit_ref_stage{stage}.pt"), weights_only=False)
        kw = dict(d["kwargs"], vit_blk=m.ImageCondDiTBlockPixelArtRMSNormClayLRM)
        with contextlib.redirect_stdout(io.StringIO()):
            model = (m.DiT_I23D_PCD_PixelArt_noclip(**kw) if stage == 1 else m.DiT_I23D_PCD_PixelArt_noclip_clay_stage2(use_pe_cond=True, **kw))
        model.load_state_dict(d["state_dict"])
        return model.eval(), d

    def engine_sample(model, cond, uc, batch_size, shape, cfg_scale, seed, method, num_steps, keep=None):
        # flow_matching_trainer.py:700-744, line by line (dist_util.dev() = cpu here; self.dtype = bf16 as the release's AMP;
        # the autocast context is a no-op for the fp32 golden model; triplane_scaling_divider = 1)
        sample_fn = sampler.sample_ode(sampling_method=method, num_steps=num_steps, cfg=True)
        torch.manual_seed(seed)
        zs = torch.randn(batch_size, *shape).to(torch.bfloat16)
        c_out = {k: torch.cat((cond[k], uc[k]), 0) for k in cond}
        zs = torch.cat([zs, zs], 0)
        traj = sample_fn(zs.float(), model.forward_with_cfg, context=c_out, cfg_scale=cfg_scale)
        if keep is not None:
            keep.append(traj.clone())
        samples, _ = traj[-1].chunk(2, dim=0)
        return samples

    out = {}
    with torch.no_grad():
        m1, d1 = load(1)
        m2, d2 = load(2)
        S, Ltok = 2, d1["x"].shape[1]
        cond = {k: d1["context"][k][:S].clone() for k in ("img_crossattn", "img_vector")}
        uc = {k: torch.zeros_like(v) for k, v in cond.items()}       # get_unconditional_conditioning with force-zero ucg
        for method, steps in (("euler", 7), ("dopri5", 5)):
            keep = []
            xyz = engine_sample(m1, cond, uc, S, (Ltok, 3), 4.0, 42, method, steps, keep)
            fps_xyz = (xyz * 0.164).clip(-0.45, 0.45)                # :987-1000 xyz_std, :1079 clip
            cond2 = dict(cond)
            cond2["fps-xyz"] = fps_xyz / 0.45                        # sgm PCD_Scaler (modules.py:1746-1768)
            uc2 = dict(cond2)                                        # stage 2: ucg_keys match no input key -> uc == c (:1039,1148-1155)
            latent = engine_sample(m2, cond2, uc2, S, (Ltok, 10), 4.0, 42, method, steps, keep)
            out[method] = dict(num_steps=steps, xyz=xyz, fps_xyz=fps_xyz, latent=latent, traj1=keep[0], traj2=keep[1], nfe_last=nfe["n"])
            print(method, "xyz", float(xyz.abs().mean()), "latent", float(latent.abs().mean()), "nfe", nfe["n"])
    out["cond"] = cond
    for k in [k for k in sys.modules if k == "transport" or k.startswith("transport.")]:
        del sys.modules[k]
    for k, v in saved.items():
        if v is not None:
            sys.modules[k] = v
        else:
            sys.modules.pop(k, None)
    return out


if __name__ == "__main__":
    torch.save(dict(render=make_render(), sample=make_sample()), os.path.join(HERE, "loop_ref.pt"))
    print("loop_ref.pt", os.path.getsize(os.path.join(HERE, "loop_ref.pt")))
