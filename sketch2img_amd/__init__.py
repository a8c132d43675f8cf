"""sketch2img_amd - MI355X (gfx950) native hot path of the sketch-guided diffusion sampler.

Compute lives in ``libskg.so`` (hand-written HIP, C ABI in include/skg.h); this package is the
Python host side: weight packing, the UNet / LGP graphs, the sampling loop and the mirror of the
reference's ``modules.*`` interface.  Importing it without the built library raises.
"""
from . import _lib  # noqa: F401  (fails loudly if libskg.so is missing)
from .config import SD15, SD21, TINY, UNetConfig  # noqa: F401
