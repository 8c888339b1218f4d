"""scp_amd.soft_renderer -- drop-in for `import soft_renderer as sr` on the training hot path
(reference package: third-party/softras/soft_renderer/__init__.py:1-10)."""
from . import cuda, functional
from .mesh import Mesh
from .renderer import Lighting, LookAt, SoftRasterizer, SoftRenderer, Transform

__all__ = ["Mesh", "SoftRenderer", "SoftRasterizer", "Lighting", "LookAt", "Transform", "functional", "cuda"]
