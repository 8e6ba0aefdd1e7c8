"""Parameter containers mirroring rqvae/models/rqvae/layers.py of the reference (ResnetBlock :60-98,
AttnBlock :130-155, Upsample :20-29, Downsample :38-48, Normalize :16-17).  They define the
state_dict names/shapes only; the arithmetic of their forward passes runs inside librqamd's
encoder/decoder engine (csrc/engine_vae.hip), so calling them directly is not supported."""
import torch
from torch import nn


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f'{type(self).__name__} is a parameter container; run the model through '
                           'RQVAE.encode / decode / decode_code (librqamd engine)')


class Upsample(_Holder):
    """nearest x2 (+ a 3 x 3 conv when with_conv; without it the engine runs the bare upsample: rqamd_vae_set_option)"""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)


class Downsample(_Holder):
    """3 x 3 / stride-2 conv behind a (0, 1, 0, 1) zero pad when with_conv, else a 2 x 2 average pool"""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)


class ResnetBlock(_Holder):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        if conv_shortcut or temb_channels > 0:
            raise NotImplementedError('conv_shortcut / temb are never used by the RQ-VAE configs')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.checkpointing = False
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout, inplace=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)


class AttnBlock(_Holder):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)
