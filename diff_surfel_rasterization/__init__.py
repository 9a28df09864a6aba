"""Import shim: lets the reference's ``from diff_surfel_rasterization import GaussianRasterizationSettings,
GaussianRasterizer`` (/root/reference/nsr/gs_surfel.py:15) resolve to the MI355X-native implementation when this
repository root is on ``sys.path`` -- no change to the reference's scripts."""
from gaussiananything_amd.diff_surfel_rasterization import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_views,
)
