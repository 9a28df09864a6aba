#!/usr/bin/env python3
"""Generates tests/golden/dit_ref_stage{1,2}.pt from the REFERENCE'S OWN DiT classes (imported from /root/reference
through tests/golden/ref_dit_loader.py).  Run once in the build container:  python tests/golden/make_dit_golden.py

Each file holds: the constructor arguments, the reference state_dict (seeded init, zero-initialised tensors re-drawn),
seeded inputs, and the fp32 outputs of ``model.forward`` and ``model.forward_with_cfg`` computed by the reference code
on CPU.  Small shapes (hidden 128, 2 heads of 64, depth 3, 48 tokens, 40 context tokens of width 64) keep the
fixtures at a few MB; the arithmetic path is the one the release models take
(DiT_I23D_PCD_PixelArt_noclip[+_clay_stage2] with ImageCondDiTBlockPixelArtRMSNormClayLRM blocks).
"""
import contextlib
import io
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_dit_loader as L  # noqa: E402


def make(stage):
    m = L.install()
    torch.manual_seed(100 + stage)
    kw = dict(input_size=8, patch_size=1, in_channels=3 if stage == 1 else 10, hidden_size=128, depth=3, num_heads=2,
              num_classes=0, learn_sigma=False, context_dim=64, pooling_ctx_dim=64, roll_out=True,
              vit_blk=m.ImageCondDiTBlockPixelArtRMSNormClayLRM, use_clay_ca=True)
    with contextlib.redirect_stdout(io.StringIO()):
        if stage == 1:
            model = m.DiT_I23D_PCD_PixelArt_noclip(**kw)
        else:
            model = m.DiT_I23D_PCD_PixelArt_noclip_clay_stage2(use_pe_cond=True, **kw)
    L.rerandomize_zero_init(model, seed=7 + stage)
    model.eval()
    g = torch.Generator().manual_seed(stage)
    B, Ltok, M = 4, 48, 40
    x = torch.randn(B, Ltok, kw["in_channels"], generator=g)
    t = torch.tensor([0.37, 0.37, 0.37, 0.37])
    ctx = {"img_crossattn": torch.randn(B, M, 64, generator=g), "img_vector": torch.randn(B, 64, generator=g)}
    ctx["img_crossattn"][B // 2:] = 0   # the unconditional half of a CFG batch is all zeros (sgm conditioner)
    ctx["img_vector"][B // 2:] = 0
    if stage == 2:
        ctx["fps-xyz"] = (torch.rand(B, Ltok, 3, generator=g) - 0.5) * 0.9
    with torch.no_grad():
        y = model(x, t, ctx)
        y_cfg = model.forward_with_cfg(x, t, ctx, 4.0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    kw_save = {k: v for k, v in kw.items() if k != "vit_blk"}
    torch.save(dict(kwargs=kw_save, state_dict=sd, x=x, t=t, context=ctx, y=y, y_cfg=y_cfg, cfg_scale=4.0),
               os.path.join(HERE, f"dit_ref_stage{stage}.pt"))
    print(f"stage {stage}: |y| mean {y.abs().mean():.4f}, params {sum(v.numel() for v in sd.values())}")


if __name__ == "__main__":
    make(1)
    make(2)
    for f in sorted(os.listdir(HERE)):
        if f.startswith("dit_ref"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
