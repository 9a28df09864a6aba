"""MI355X-native DiT denoisers of the cascaded image-to-3D sampler (same class names, constructor arguments,
``forward`` / ``forward_with_cfg`` surfaces and state-dict keys as /root/reference/dit/dit_i23d.py)."""
from .dit_i23d import (  # noqa: F401
    DiT_I23D_PCD_PixelArt_noclip,
    DiT_I23D_PCD_PixelArt_noclip_clay_stage2,
    DiT_models,
)
