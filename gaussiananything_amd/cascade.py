"""The cascaded image-to-3D sampling loop around the hot path: what ``FlowMatchingEngine.sample`` and the two sampling
scripts do between the conditioner and the renderer (/root/reference/nsr/lsgm/flow_matching_trainer.py:700-744 ``sample``;
:1206-1225 stage hand-off; :1400-1424 ``render_gs_video_given_latent``; shell_scripts/release/inference/i23d/*.sh).

    stage 1  z ~ N(0, I) [S, 768, 3]  --250-step ODE, CFG-->  normalised point cloud;  x 0.164 (``xyz_std``, :987-1000), clipped to
             +-0.45 when it is read back as the stage-2 condition (:1079)
    stage 2  z ~ N(0, I) [S, 768, 10], context + {'fps-xyz'}  --250-step ODE, CFG-->  KL latent
    decode   {latent_normalized, query_pcd_xyz} -> surfels (SurfelDecoder)  ->  triplane_decode -> renders per level

The reference writes the stage-1 cloud to a PLY file and a second script loads it; here the tensor stays on the device.
Host orchestration only -- every kernel launch is behind the three C-ABI headers.
"""
from __future__ import annotations

import torch

from .transport import Sampler, create_transport

XYZ_STD = 0.164  # flow_matching_trainer.py:987
PCD_SCALING_FACTOR = 0.45  # sgm/configs/stage2-i23d.yaml:55-57 -> PCD_Scaler (sgm/modules/encoders/modules.py:1746-1768)


@torch.no_grad()
def sample(model, cond, uc, shape, batch_size=1, cfg_scale=4.0, seed=42, num_steps=250, sampling_method="dopri5",
           transport_sampler=None, noise_dtype=torch.bfloat16, stats=None, dedup_noop_cfg=True, **ode_kwargs):
    """``FlowMatchingEngine.sample`` (flow_matching_trainer.py:700-744): CPU-seeded noise, CFG batch = [cond | uncond],
    ``sample_ode(num_steps=250, cfg=True)`` (dopri5 by default, as upstream), last state, conditional half.

    ``noise_dtype``: the reference draws fp32 noise on the CPU and rounds it to the engine dtype before sampling
    (``.to(self.dtype)``, :720; the release runs bf16 AMP), so the initial state is a bf16-representable tensor; the ODE
    state is fp32 from the first update on.  ``None`` keeps the un-rounded fp32 draw.

    ``dedup_noop_cfg``: when the unconditional conditioning IS the conditional one (``uc[k] is cond[k]`` for every key -- the
    release's stage 2, ``stage2_conditioning``), both halves of the CFG batch are the same sequence, ``forward_with_cfg``
    returns ``uncond + s * (cond - uncond) = cond`` and the two halves of the ODE state stay equal (dopri5's RMS norm over
    the doubled state equals that over one half): the denoiser is evaluated on the conditional half alone -- the same
    numbers up to the GEMM tile shapes chosen for the smaller batch, half the work."""
    if transport_sampler is None:
        transport_sampler = Sampler(create_transport("GVP", "velocity", None, None, None, snr_type="uniform"))
    sample_fn = transport_sampler.sample_ode(sampling_method=sampling_method, num_steps=num_steps, cfg=True, **ode_kwargs)
    dev = next(model.parameters()).device
    torch.manual_seed(seed)
    zs = torch.randn(batch_size, *shape).to(dev)
    if noise_dtype is not None:
        zs = zs.to(noise_dtype).float()
    if dedup_noop_cfg and all(uc[k] is cond[k] for k in cond):
        def cond_only(x, t, context=None, cfg_scale=None):
            return model.forward(x, t, context)
        samples = sample_fn(zs, getattr(model, "forward_cond", cond_only), context=dict(cond), cfg_scale=cfg_scale)[-1]
        if stats is not None:
            stats.update(getattr(getattr(transport_sampler, "last_ode", None), "last_stats", {}) or {}, noop_cfg_dedup=True)
        return samples
    c_out = {k: torch.cat((cond[k], uc[k]), 0) for k in cond}
    zs = torch.cat([zs, zs], 0)
    samples = sample_fn(zs, model.forward_with_cfg, context=c_out, cfg_scale=cfg_scale)[-1]
    if stats is not None:   # function evaluations / accepted / rejected steps of this stage's ODE solve
        stats.update(getattr(getattr(transport_sampler, "last_ode", None), "last_stats", {}) or {})
    samples, _ = samples.chunk(2, dim=0)
    return samples


@torch.no_grad()
def condition_on_image(embedder, image):
    """The conditioning dicts of the i23d release from an image batch in [-1, 1] (``FrozenDinov2ImageEmbedder`` with
    ``output_cls=True``, configs: sgm/modules/encoders/modules.py:791-931): cond = {'img_crossattn': patch tokens
    [S,1369,1024], 'img_vector': cls token [S,1024]}; the unconditional half of CFG is all zeros
    (flow_matching_trainer.py:1148-1153 ``get_unconditional_conditioning`` with ucg force-zero)."""
    tokens, cls = embedder(image, no_dropout=True)  # get_unconditional_conditioning forces ucg_rate = 0 (modules.py:185-194)
    cond = {"img_crossattn": tokens.contiguous(), "img_vector": cls.contiguous()}
    return cond, {k: torch.zeros_like(v) for k, v in cond.items()}


@torch.no_grad()
def stage2_conditioning(cond, uc, fps_xyz, zero_image_uc=False):
    """Stage-2 conditioning dicts from the stage-1 cloud, as the release's conditioner builds them
    (sgm/configs/stage2-i23d.yaml): the ``fps-xyz`` embedder is ``PCD_Scaler`` -- the denoiser's XYZPosEmbed sees
    ``xyz / 0.45`` (only the decoder gets the raw cloud).  Stage 2 runs with ``cond_key = 'img-xyz'``, so
    ``ucg_keys = ['img-xyz']`` matches no embedder input key ('img', 'fps-xyz') and ``get_unconditional_conditioning``
    returns uc == c (flow_matching_trainer.py:1039,1148-1155): classifier-free guidance is a no-op there.
    ``zero_image_uc=True`` is the non-reference variant that guides against the zero-image branch."""
    scaled = fps_xyz / PCD_SCALING_FACTOR
    cond2 = dict(cond)
    cond2["fps-xyz"] = scaled
    uc2 = dict(uc) if zero_image_uc else dict(cond)
    uc2["fps-xyz"] = scaled
    return cond2, uc2


@torch.no_grad()
def cascade(stage1, stage2, decoder, cond, uc, cameras=None, cfg_scale=4.0, seed=42, num_steps=250,
            sampling_method="dopri5", render_all_scale=True, stage2_zero_image_uc=False, stats=None, **ode_kwargs):
    """Stage 1 -> stage 2 -> surfel decode (-> renders when ``cameras`` = {cam_view, cam_view_proj [B,V,4,4], cam_pos
    [B,V,3], tanfov} is given).  ``cond`` / ``uc``: {'img_crossattn' [S,1369,1024], 'img_vector' [S,1024]}."""
    S = cond["img_crossattn"].shape[0]
    L = decoder.vit_decoder.pos_embed.shape[1]  # 768 latent tokens in the release (z_shape, flow_matching_trainer.py:1158)
    st1, st2 = ({}, {}) if stats is not None else (None, None)
    xyz = sample(stage1, cond, uc, (L, stage1.in_channels), S, cfg_scale, seed, num_steps, sampling_method, stats=st1,
                 **ode_kwargs)
    fps_xyz = (xyz * XYZ_STD).clip(-0.45, 0.45)
    cond2, uc2 = stage2_conditioning(cond, uc, fps_xyz, zero_image_uc=stage2_zero_image_uc)
    latent = sample(stage2, cond2, uc2, (L, stage2.in_channels), S, cfg_scale, seed, num_steps, sampling_method,
                    stats=st2, **ode_kwargs)
    if stats is not None:
        stats.update(stage1=st1, stage2=st2)
    ret = decoder.decode(latent, fps_xyz)
    if cameras is not None:
        ret["renders"] = decoder.triplane_decode(ret, cameras, render_all_scale=render_all_scale)
    return ret
