#!/usr/bin/env python3
"""Round-6 parity evidence at the stated configurations (GPU box; minutes of host time, so not in pytest).

  traj250   BASELINE configs[2] AS WRITTEN: DiT-PixArt-PCD-CLAY-B (depth 12), guided Euler over 250 grid points = 249 evaluations,
            HIP path (bf16 MFMA operands, fused on-device step) against oracle/trajectory.py (fp32 model, fp64 integrator):
            relative L2 at every 10th saved state, the end state and the worst state.
  cascade   BASELINE configs[3] at release size with 25 grid points per stage: stage-1 DiT-L (CFG) -> x 0.164 / clip -> stage-2
            DiT-L (uc == c) -> surfel decode -> 8 x 512^2 renders of gaussians_upsampled_3, HIP against an all-fp32 oracle cascade
            (oracle/dit.py + oracle/ode.py + oracle/decode.py + oracle/surfel.py; flow_matching_trainer.py:700-744, 1206-1225).
            FREE-RUNNING errors (each HIP stage fed by the HIP stage before it) and STAGE-WISE errors (each HIP stage fed the
            ORACLE's input), so that a divergence can be attributed to the stage that causes it.

usage (GPU box): python tools/parity_r6.py traj250|cascade [--points N] > gpurun_out/r6_*.txt
The oracle is only the checker here (tools/, like tests/)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def host_threads():
    return min(64, os.cpu_count() or 1)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def traj250(points):
    from gaussiananything_amd.transport import Sampler, create_transport
    from oracle import trajectory as otraj
    dev = torch.device("cuda:0")
    model, sd = otraj.release_model("DiT-PixArt-PCD-CLAY-B", 3)
    x0, ctx = otraj.release_inputs(3, cfg=True)
    model.to(dev)
    sampler = Sampler(create_transport("GVP", "velocity", None, None, None, snr_type="uniform"))
    fn = sampler.sample_ode(sampling_method="euler", num_steps=points)
    with torch.no_grad():
        out = fn(x0.to(dev), model.forward_with_cfg, context={k: v.to(dev) for k, v in ctx.items()}, cfg_scale=4.0)
    torch.cuda.synchronize()
    got = out.double().cpu().numpy()
    t0 = time.perf_counter()
    ref = otraj.integrate(sd, x0, ctx, 4.0, "euler", points, cfg=True, threads=host_threads())
    cpu_s = time.perf_counter() - t0
    errs = [rel(got[i], ref[i]) for i in range(1, points)]
    print(f"# BASELINE configs[2] as written: DiT-PixArt-PCD-CLAY-B depth 12, CFG batch 2 x 768 tokens x 3, scale 4, euler, {points} grid points "
          f"= {points - 1} evaluations")
    print("# HIP (bf16 MFMA operands, fp32 accumulate / residual / state, fused on-device step) vs oracle/trajectory.py (fp32 model, fp64 integrator)")
    print(f"# oracle: {cpu_s:.0f} s on {host_threads()} host threads; state moved by rel. L2 {rel(ref[-1], ref[0]):.3f} from the start state")
    print("grid_point  t        rel_l2(HIP, oracle)")
    for i in list(range(10, points - 1, 10)) + [points - 1]:
        print(f"{i:10d}  {i / (points - 1):.4f}  {errs[i - 1]:.4e}")
    worst = int(np.argmax(errs)) + 1
    print(f"end_state_rel_l2   {errs[-1]:.4e}")
    print(f"worst_state_rel_l2 {max(errs):.4e} at grid point {worst}")
    print(f"bar (tests/test_dit_gpu.py, 25 grid points): 3e-2 -> {'within' if errs[-1] < 3e-2 else 'BEYOND'} at {points} points")


def cascade(points):
    import bench
    from gaussiananything_amd import cascade as gc, synthetic
    from gaussiananything_amd.transport import Sampler, create_transport
    from oracle import decode as odec, surfel as osurf, trajectory as otraj
    dev = torch.device("cuda:0")
    torch.set_num_threads(host_threads())
    m1, m2, dec = bench.build_cascade_models(dev)
    sd1 = {k: v.detach().float().cpu() for k, v in m1.state_dict().items()}
    sd2 = {k: v.detach().float().cpu() for k, v in m2.state_dict().items()}
    sdd = {k: v.detach().float().cpu() for k, v in dec.state_dict().items()}
    cams = synthetic.eval_cameras(8)
    c = {"cam_view": cams["cam_view"][None].to(dev), "cam_view_proj": cams["cam_view_proj"][None].to(dev),
         "cam_pos": cams["cam_pos"][None].to(dev), "tanfov": cams["tanfov"]}
    g = torch.Generator().manual_seed(1000)
    cond = {"img_crossattn": torch.randn(1, 1369, 1024, generator=g), "img_vector": torch.randn(1, 1024, generator=g)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    cfg_scale, seed, L = 4.0, 42, 768
    sampler = Sampler(create_transport("GVP", "velocity", None, None, None, snr_type="uniform"))

    def noise(C):   # FlowMatchingEngine.sample: CPU-seeded noise rounded to the engine dtype (flow_matching_trainer.py:718-720)
        torch.manual_seed(seed)
        return torch.randn(1, L, C).to(torch.bfloat16).float()

    # ---------------- oracle cascade (fp32 models, fp64 integrator, fp32 decode, C raster oracle) -----------------------------
    t0 = time.perf_counter()
    z1 = noise(3)
    ctx1 = {k: torch.cat([cond[k], uc[k]], 0) for k in cond}
    o_xyz = torch.from_numpy(otraj.integrate(sd1, torch.cat([z1, z1], 0), ctx1, cfg_scale, "euler", points, cfg=True)[-1][:1]).float()
    o_fps = (o_xyz * gc.XYZ_STD).clip(-0.45, 0.45)
    t1 = time.perf_counter()

    def oracle_stage2(fps):
        ctx2 = dict(cond)
        ctx2["fps-xyz"] = fps / gc.PCD_SCALING_FACTOR
        return torch.from_numpy(otraj.integrate(sd2, noise(10), ctx2, cfg_scale, "euler", points, cfg=False)[-1]).float()
    o_lat = oracle_stage2(o_fps)
    t2 = time.perf_counter()
    o_dec = odec.decode(sdd, o_lat, o_fps)
    o_surf = o_dec["gaussians_upsampled_3"]
    t3 = time.perf_counter()

    def oracle_render(surf):
        m, o, s, r, col = synthetic.split_gaussians(surf[0])
        outs = [osurf.rasterize(m.numpy(), o.numpy(), col.numpy(), s.numpy(), r.numpy(), cams["cam_view"][v].numpy(),
                                cams["cam_view_proj"][v].numpy(), np.ones(3, np.float32), 512, 512) for v in range(8)]
        return np.stack([np.concatenate([q["color"], q["allmap"]], 0) for q in outs], 0), sum(q["D"] for q in outs)
    o_img, o_D = oracle_render(o_surf)
    t4 = time.perf_counter()

    # ---------------- HIP cascade, free running --------------------------------------------------------------------------------
    cond_d = {k: v.to(dev) for k, v in cond.items()}
    uc_d = {k: v.to(dev) for k, v in uc.items()}

    def hip_stage1():
        return gc.sample(m1, cond_d, uc_d, (L, 3), 1, cfg_scale, seed, points, "euler", transport_sampler=sampler)

    def hip_stage2(fps_dev):
        c2, u2 = gc.stage2_conditioning(cond_d, uc_d, fps_dev)
        return gc.sample(m2, c2, u2, (L, 10), 1, cfg_scale, seed, points, "euler", transport_sampler=sampler)

    def hip_render(surf_dev):
        from gaussiananything_amd.diff_surfel_rasterization import rasterize_views
        gdev = surf_dev[0].contiguous().float()
        color, _radii, allmap, ws = rasterize_views(gdev[:, 0:3], gdev[:, 3:4], gdev[:, 10:13], gdev[:, 4:6], gdev[:, 6:10], c["cam_view"][0],
                                                    c["cam_view_proj"][0], torch.ones(3, device=dev), 512, 512, 1.0)
        torch.cuda.synchronize()
        return torch.cat([color, allmap], 1).cpu().numpy(), int(ws.status().cpu()[0])

    with torch.no_grad():
        h_xyz = hip_stage1()
        h_fps = (h_xyz * gc.XYZ_STD).clip(-0.45, 0.45)
        h_lat = hip_stage2(h_fps)
        h_surf = dec.decode(h_lat, h_fps)["gaussians_upsampled_3"]
        h_img, h_D = hip_render(h_surf)
        # stage-wise: every HIP stage on the ORACLE's input
        s_lat = hip_stage2(o_fps.to(dev))
        s_surf = dec.decode(o_lat.to(dev), o_fps.to(dev))["gaussians_upsampled_3"]
        s_img, s_D = hip_render(o_surf.to(dev))
    torch.cuda.synchronize()

    names = ["R", "G", "B", "depth", "alpha", "nx", "ny", "nz", "median_depth", "distortion"]
    groups = {"xyz": slice(0, 3), "opacity": slice(3, 4), "scale": slice(4, 6), "rotation": slice(6, 10), "rgb": slice(10, 13)}

    def pixel_table(a, b):
        rows = []
        for ch, nm in enumerate(names):
            d = a[:, ch].astype(np.float64) - b[:, ch].astype(np.float64)
            rows.append((nm, float((d * d).mean()), float(np.abs(d).max()), int((np.abs(d) > 1e-4).sum())))
        return rows

    def show_pixels(title, a, b):
        print(title)
        print("  channel        MSE          max_abs      pixels>1e-4 (of %d)" % (a.shape[0] * a.shape[2] * a.shape[3]))
        for nm, mse, mx, cnt in pixel_table(a, b):
            print(f"  {nm:13s}  {mse:.3e}    {mx:.3e}    {cnt}")

    def show_surfels(title, a, b):
        a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
        print(title + "  all 13: %.3e" % rel(a, b) + "   " + "  ".join(f"{k} {rel(a[..., s], b[..., s]):.3e}" for k, s in groups.items()))

    print(f"# BASELINE configs[3] at release size: DiT-L (CFG batch 2, scale {cfg_scale}) -> x{gc.XYZ_STD} / clip 0.45 -> stage-2 DiT-L (uc == c, batch 1) -> "
          f"SurfelDecoder -> 73 728 surfels -> 8 x 512^2; euler, {points} grid points = {points - 1} evaluations per stage; seeded random weights "
          "(bench.build_cascade_models), conditioning seed 1000, noise seed 42 rounded to bf16")
    print(f"# oracle cascade on {host_threads()} host threads: stage 1 {t1 - t0:.0f} s, stage 2 {t2 - t1:.0f} s, decode {t3 - t2:.0f} s, raster {t4 - t3:.0f} s")
    print("\n== FREE-RUNNING (each HIP stage fed by the HIP stage before it) vs the oracle cascade ==")
    print(f"stage 1  xyz (normalised point cloud)      rel_l2 {rel(h_xyz.cpu().numpy(), o_xyz.numpy()):.3e}   max_abs {float((h_xyz.cpu() - o_xyz).abs().max()):.3e}"
          f"   (|xyz| rms {float(o_xyz.pow(2).mean().sqrt()):.3f})")
    print(f"hand-off fps_xyz = clip(0.164 xyz, +-0.45)   rel_l2 {rel(h_fps.cpu().numpy(), o_fps.numpy()):.3e}   clipped coordinates: oracle {int((o_fps.abs() >= 0.45).sum())}, HIP {int((h_fps.abs() >= 0.45).sum())}")
    print(f"stage 2  latent [1,768,10]                 rel_l2 {rel(h_lat.cpu().numpy(), o_lat.numpy()):.3e}   max_abs {float((h_lat.cpu() - o_lat).abs().max()):.3e}")
    show_surfels("decode   surfels [1,73728,13]               rel_l2", h_surf, o_surf)
    print(f"raster   num_rendered D: oracle {o_D}, HIP {h_D}")
    show_pixels("raster   8 x 512^2 renders, HIP cascade vs oracle cascade:", h_img, o_img)
    print("\n== STAGE-WISE (each HIP stage fed the ORACLE's input): which stage moves the result ==")
    print(f"stage 2 on the oracle's fps_xyz             rel_l2 {rel(s_lat.cpu().numpy(), o_lat.numpy()):.3e}")
    show_surfels("decode on the oracle's latent + fps_xyz    rel_l2", s_surf, o_surf)
    print(f"raster on the oracle's surfels: D oracle {o_D}, HIP {s_D} ({'bins agree in size' if s_D == o_D else 'DIFFERENT'})")
    show_pixels("raster on the oracle's surfels, HIP vs oracle:", s_img, o_img)
    show_pixels("raster of the HIP decode of the ORACLE's latent (decode + raster only) vs oracle:", hip_render(s_surf)[0], o_img)
    # ---------------- the same question from the other side: the ORACLE's stages on the HIP cascade's inputs -----------------------
    # (if the free-running divergence is the models' sensitivity to their inputs and not an error of a HIP stage, every HIP stage
    #  must agree with the oracle stage evaluated on the SAME input)
    t5 = time.perf_counter()
    c_lat = oracle_stage2(h_fps.cpu())
    c_surf = odec.decode(sdd, h_lat.cpu(), h_fps.cpu())["gaussians_upsampled_3"]
    c_img, c_D = oracle_render(h_surf.cpu())
    print(f"\n== ON THE HIP CASCADE'S OWN INTERMEDIATES (oracle stage on the HIP stage's input; {time.perf_counter() - t5:.0f} s) ==")
    print(f"stage 2: oracle(HIP fps_xyz) vs HIP latent            rel_l2 {rel(h_lat.cpu().numpy(), c_lat.numpy()):.3e}"
          f"      [oracle(HIP fps_xyz) vs oracle(oracle fps_xyz): {rel(c_lat.numpy(), o_lat.numpy()):.3e} = the fp32 MODEL's own response to the 1e-2 hand-off difference]")
    show_surfels("decode: oracle(HIP latent, HIP fps_xyz) vs HIP surfels  rel_l2", h_surf, c_surf)
    print(f"raster: oracle(HIP surfels) vs HIP renders: D oracle {c_D}, HIP {h_D}")
    show_pixels("raster: oracle(HIP surfels) vs HIP renders:", h_img, c_img)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["traj250", "cascade"])
    ap.add_argument("--points", type=int, default=None)
    a = ap.parse_args()
    if a.what == "traj250":
        traj250(a.points or 250)
    else:
        cascade(a.points or 25)
