"""ctypes binding of libpnpadmm.so (include/pnpadmm.h).  No fallback: if the HIP library is
missing or a call fails, this raises - the product path never routes through CPU code."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNP_LIB_PATH") or os.path.join(_HERE, "csrc", "libpnpadmm.so")   # override: kernel experiments

PNP_FLAG_PROFILE = 1
PNP_FLAG_NO_DENOISER = 2
PNP_FLAG_KEEP_STAGES = 4
PNP_FLAG_BF16_CONVS = 8
PNP_FLAG_PROFILE_LAYERS = 16
PROFILE_CLASSES = 6
PROFILE_CLASS_NAMES = ("conv3x3_mfma", "conv_first", "conv_last", "fft_rows", "fft_cols_prox", "other")
N_LAYERS = 28


class pnp_config(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32)]


_vp, _fp, _u8p = C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers

# name -> (restype, argtypes); every symbol include/pnpadmm.h declares
SIGNATURES = {
    "pnp_create": (C.c_int, [C.POINTER(pnp_config), C.POINTER(C.c_void_p)]),
    "pnp_destroy": (C.c_int, [C.c_void_p]),
    "pnp_last_error": (C.c_char_p, []),
    "pnp_version": (C.c_char_p, []),
    "pnp_load_unet_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "pnp_reset": (C.c_int, [C.c_void_p, _fp, _fp, _u8p, C.c_int, _fp, _fp, _fp, _vp]),
    "pnp_set_kspace": (C.c_int, [C.c_void_p, _fp, _u8p, C.c_int, _vp]),
    "pnp_step": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _u8p, _vp]),
    "pnp_denoise": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _vp]),
    "pnp_fft2c": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "pnp_prox_dual": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _fp, _vp]),
    "pnp_psnr": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _vp]),
    "pnp_snapshot_bytes": (C.c_size_t, [C.c_void_p]),
    "pnp_snapshot": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp, _vp, _vp]),
    "pnp_restore": (C.c_int, [C.c_void_p, _vp, _fp, _fp, _fp, _fp, _vp]),
    "pnp_unet_read_stage": (C.c_int, [C.c_void_p, C.c_int, _fp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), _vp]),
    "pnp_profile_reset": (C.c_int, [C.c_void_p]),
    "pnp_profile_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pnp_profile_layers": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pnp_conv_algorithms": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "pnp_bf16_weight_terms": (C.c_int, [C.c_void_p]),
    "pnp_workspace_bytes": (C.c_size_t, [C.c_void_p]),
}

_lib = None


class PnPError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the shared library (once).  Raises PnPError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PnPError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C {os.path.dirname(LIB_PATH)}`; there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pnp_last_error()
        raise PnPError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
