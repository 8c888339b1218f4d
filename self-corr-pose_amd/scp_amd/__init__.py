"""scp_amd -- MI355X-native hot path of kywind/self-corr-pose behind the reference's own Python API.

Sub-packages / modules
  capi            ctypes binding of lib/libscp_hip.so (the C ABI of include/scp_hip.h)
  soft_renderer   drop-in for the SoftRas surface the reference uses (`import soft_renderer as sr`)
  install()       registers the drop-ins under the reference's module names
"""
import sys

__all__ = ["install"]


def install():
    """Make `import soft_renderer` (and soft_renderer.cuda.soft_rasterize) resolve to this
    package, the way the reference's `python setup.py install` of third-party/softras would."""
    from . import soft_renderer as sr
    from .soft_renderer import cuda as sr_cuda
    from .soft_renderer import functional as srf
    from .soft_renderer.cuda import soft_rasterize as native
    sys.modules["soft_renderer"] = sr
    sys.modules["soft_renderer.functional"] = srf
    sys.modules["soft_renderer.cuda"] = sr_cuda
    sys.modules["soft_renderer.cuda.soft_rasterize"] = native
    return sr
