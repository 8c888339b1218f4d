"""scp_amd.soft_renderer.cuda -- stands where the reference's compiled extension package does."""
from . import soft_rasterize  # noqa: F401
