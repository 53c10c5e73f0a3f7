"""MI355X-native PnP-ADMM CS-MRI engine: the hot path of joesharratt1229/DT4Image_Restoration's
`evaluation/env.py` as HIP kernels behind a C ABI (include/pnpadmm.h).  Importing this package does not
need a GPU; constructing an engine does."""
from .unet_spec import UNET_LAYERS, STATE_DICT_KEYS, FLOPS_PER_PIXEL  # noqa: F401

__all__ = ["PnPEnv", "UNetDenoiser2D", "PnPEngine", "fft", "ifft"]


def __getattr__(name):
    if name == "PnPEnv":
        from .env import PnPEnv
        return PnPEnv
    if name == "UNetDenoiser2D":
        from .denoiser import UNetDenoiser2D
        return UNetDenoiser2D
    if name == "PnPEngine":
        from .engine import PnPEngine
        return PnPEngine
    if name in ("fft", "ifft"):
        from . import transformations
        return getattr(transformations, name)
    raise AttributeError(name)
