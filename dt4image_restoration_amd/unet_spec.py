"""Static description of the plug-in denoiser the PnP-ADMM hot path runs.

The layer table restates the architecture of the reference U-Net
(/root/reference/evaluation/noise.py:101-113 `UNet.__init__`, `ConvBlock` :88-98,
`ConvLayer` :75-85, `outconv` :64-71) as data: 27 conv3x3(s1,p1,bias)+LeakyReLU(0.2)
layers in 9 stages of 3, followed by one conv1x1.  The state_dict key names are the
ones `UNetDenoiser2D.__init__` loads (noise.py:146-148), so a checkpoint written for
the reference is ingested unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

LEAKY_SLOPE = 0.2  # noise.py:89  act=nn.LeakyReLU(0.2)

# input transform applied in front of a stage's first conv
SRC_PLAIN = 0      # previous tensor as is
SRC_SIGMA = 1      # image channel + constant sigma plane   (noise.py:159-162)
SRC_POOL = 2       # MaxPool2d(2) of previous tensor         (noise.py:22-25)
SRC_UPCAT = 3      # cat([skip, bilinear_up2(prev)], dim=1)  (noise.py:46-59)


@dataclass(frozen=True)
class ConvSpec:
    index: int        # 0..27 in execution order
    key: str          # state_dict prefix; "<key>.weight" / "<key>.bias"
    cin: int
    cout: int
    ksize: int        # 3 or 1
    level: int        # resolution level: spatial = (H >> level, W >> level)
    src: int          # SRC_* of the input of this conv
    cskip: int = 0    # SRC_UPCAT only: channels taken from the skip tensor (first in cat order)

    @property
    def weight_key(self) -> str:
        return self.key + ".weight"

    @property
    def bias_key(self) -> str:
        return self.key + ".bias"

    @property
    def macs_per_out_pixel(self) -> int:
        return self.cin * self.cout * self.ksize * self.ksize


def _stage(idx: int, prefix: str, cin: int, cout: int, level: int, src: int, cskip: int = 0) -> List[ConvSpec]:
    out = []
    for j in range(3):
        out.append(ConvSpec(index=idx + j,
                            key=f"{prefix}.conv-{j}.conv2d",
                            cin=cin if j == 0 else cout, cout=cout, ksize=3, level=level,
                            src=src if j == 0 else SRC_PLAIN,
                            cskip=cskip if j == 0 else 0))
    return out


def unet_layers() -> List[ConvSpec]:
    """The 28 convolutions of UNet(2, 1) in execution order (noise.py:119-133)."""
    L: List[ConvSpec] = []
    L += _stage(0, "inc.conv", 2, 32, 0, SRC_SIGMA)
    L += _stage(3, "down1.mpconv.1", 32, 64, 1, SRC_POOL)
    L += _stage(6, "down2.mpconv.1", 64, 128, 2, SRC_POOL)
    L += _stage(9, "down3.mpconv.1", 128, 256, 3, SRC_POOL)
    L += _stage(12, "down4.mpconv.1", 256, 512, 4, SRC_POOL)
    L += _stage(15, "up1.conv", 512 + 256, 256, 3, SRC_UPCAT, cskip=256)
    L += _stage(18, "up2.conv", 256 + 128, 128, 2, SRC_UPCAT, cskip=128)
    L += _stage(21, "up3.conv", 128 + 64, 64, 1, SRC_UPCAT, cskip=64)
    L += _stage(24, "up4.conv", 64 + 32, 32, 0, SRC_UPCAT, cskip=32)
    L.append(ConvSpec(index=27, key="outc.conv", cin=32, cout=1, ksize=1, level=0, src=SRC_PLAIN))
    return L


UNET_LAYERS = unet_layers()
STATE_DICT_KEYS = [k for l in UNET_LAYERS for k in (l.weight_key, l.bias_key)]  # 56 keys
N_PARAMS = sum(l.macs_per_out_pixel + l.cout for l in UNET_LAYERS)            # 11,773,857
MACS_PER_PIXEL = sum(l.macs_per_out_pixel // (4 ** l.level) for l in UNET_LAYERS)  # 295,520
FLOPS_PER_PIXEL = 2 * MACS_PER_PIXEL                                             # 591,040


def conv_flops(h: int, w: int, n: int = 1) -> int:
    """Algorithmic FLOPs (2*MAC) of one denoiser forward over n slices of h x w."""
    return n * sum(2 * l.macs_per_out_pixel * (h >> l.level) * (w >> l.level) for l in UNET_LAYERS)
