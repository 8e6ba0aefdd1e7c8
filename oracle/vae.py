"""RQ-VAE oracle (numpy, fp32, NHWC internally).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates the reference's stage-1 model:
  rqvae/models/rqvae/rqvae.py     encode :80-83, decode :85-89, get_codes :91-95,
                                  decode_code :105-109, forward :74-78
  rqvae/models/rqvae/modules.py   Encoder.forward :73-98, Decoder.forward :171-202
  rqvae/models/rqvae/layers.py    nonlinearity (SiLU) :11-13, Normalize = GroupNorm(32, eps 1e-6) :16-17,
                                  Upsample (nearest x2 + conv) :31-35, Downsample (pad (0,1,0,1) + conv s2) :50-57,
                                  ResnetBlock._forward :100-120, AttnBlock.forward :158-182
  torch.nn.Conv2d (cross-correlation, zero padding), torch.nn.GroupNorm (biased variance).
"""
import numpy as np

from . import backend
from .rq import rq_quantize, rq_embed_code


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def silu(x):
    if backend.torch_on():
        import torch.nn.functional as F
        return F.silu(_t(x)).numpy()
    return x / (1.0 + np.exp(-x))


def group_norm(x, w, b, groups=32, eps=1e-6):
    """x NHWC.  layers.py:16-17"""
    B, H, W, C = x.shape
    if backend.torch_on():
        import torch.nn.functional as F
        y = F.group_norm(_t(x).permute(0, 3, 1, 2), groups, _t(w), _t(b), eps)
        return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())
    g = x.reshape(B, H * W, groups, C // groups).astype(np.float32)
    mu = g.mean((1, 3), keepdims=True, dtype=np.float32)
    xc = g - mu
    var = (xc * xc).mean((1, 3), keepdims=True, dtype=np.float32)
    y = (xc / np.sqrt(var + np.float32(eps))).reshape(B, H, W, C)
    return (y * w + b).astype(np.float32)


def conv2d(x, w, b, stride=1, pad=(1, 1, 1, 1)):
    """x NHWC fp32, w (Cout,Cin,kh,kw) torch layout, pad = (top, bottom, left, right)."""
    B, H, W, C = x.shape
    co, ci, kh, kw = w.shape
    if backend.torch_on():
        import torch.nn.functional as F
        xt = F.pad(_t(x).permute(0, 3, 1, 2), (pad[2], pad[3], pad[0], pad[1]))
        y = F.conv2d(xt, _t(w), _t(b), stride=stride)
        return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())
    xp = np.pad(x, ((0, 0), (pad[0], pad[1]), (pad[2], pad[3]), (0, 0)))
    Ho = (H + pad[0] + pad[1] - kh) // stride + 1
    Wo = (W + pad[2] + pad[3] - kw) // stride + 1
    out = np.zeros((B * Ho * Wo, co), np.float32)
    wt = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))      # (kh,kw,ci,co): contiguous BLAS operands
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride, :]
            out += np.ascontiguousarray(patch).reshape(-1, ci) @ wt[ky, kx]
    return (out + b).reshape(B, Ho, Wo, co)


class RQVAEOracle:
    def __init__(self, hps, ddconfig, params):
        self.hps, self.dd = hps, ddconfig
        self.p = {k: np.asarray(v, np.float32) for k, v in params.items()}
        self.depth = hps['code_shape'][-1]
        shared = hps.get('shared_codebook', False)
        self.codebooks = [self.p[f'quantizer.codebooks.{0 if shared else i}.weight'][:-1]
                          for i in range(self.depth)]          # padding row dropped (quantizations.py:45)

    # ------------------------------------------------------------------ layers
    def _conv(self, name, x, stride=1, pad=None):
        w = self.p[name + '.weight']
        k = w.shape[-1]
        if pad is None:
            pad = (k // 2,) * 4
        return conv2d(x, w, self.p[name + '.bias'], stride, pad)

    def _norm(self, name, x):
        return group_norm(x, self.p[name + '.weight'], self.p[name + '.bias'])

    def _res(self, name, x):
        """layers.py:100-120"""
        h = self._conv(name + '.conv1', silu(self._norm(name + '.norm1', x)))
        h = self._conv(name + '.conv2', silu(self._norm(name + '.norm2', h)))
        if name + '.nin_shortcut.weight' in self.p:
            x = self._conv(name + '.nin_shortcut', x)
        return x + h

    def _attn(self, name, x):
        """layers.py:158-182"""
        B, H, W, C = x.shape
        h = self._norm(name + '.norm', x)
        q = self._conv(name + '.q', h).reshape(B, H * W, C)
        k = self._conv(name + '.k', h).reshape(B, H * W, C)
        v = self._conv(name + '.v', h).reshape(B, H * W, C)
        w_ = (q @ k.transpose(0, 2, 1)) * np.float32(int(C) ** (-0.5))
        w_ = w_ - w_.max(-1, keepdims=True)
        w_ = np.exp(w_)
        w_ = w_ / w_.sum(-1, keepdims=True)
        h = (w_ @ v).reshape(B, H, W, C)
        return x + self._conv(name + '.proj_out', h)

    # ------------------------------------------------------------------ encoder / decoder
    def encoder(self, x_nchw):
        """modules.py:73-98"""
        dd = self.dd
        h = self._conv('encoder.conv_in', np.transpose(np.asarray(x_nchw, np.float32), (0, 2, 3, 1)))
        nres = len(dd['ch_mult'])
        for i_level in range(nres):
            for i_block in range(dd['num_res_blocks']):
                h = self._res(f'encoder.down.{i_level}.block.{i_block}', h)
                if f'encoder.down.{i_level}.attn.{i_block}.norm.weight' in self.p:
                    h = self._attn(f'encoder.down.{i_level}.attn.{i_block}', h)
            if i_level != nres - 1:
                if dd.get('resamp_with_conv', True):
                    h = self._conv(f'encoder.down.{i_level}.downsample.conv', h, stride=2, pad=(0, 1, 0, 1))
                else:                                                # layers.py:55-56: avg_pool2d(kernel_size=2, stride=2)
                    B, H, W, C_ = h.shape
                    h = h.reshape(B, H // 2, 2, W // 2, 2, C_).mean(axis=(2, 4), dtype=np.float32)
        h = self._res('encoder.mid.block_1', h)
        h = self._attn('encoder.mid.attn_1', h)
        h = self._res('encoder.mid.block_2', h)
        h = silu(self._norm('encoder.norm_out', h))
        return self._conv('encoder.conv_out', h)

    def decoder(self, z_nhwc):
        """modules.py:171-202"""
        dd = self.dd
        h = self._conv('decoder.conv_in', z_nhwc)
        h = self._res('decoder.mid.block_1', h)
        h = self._attn('decoder.mid.attn_1', h)
        h = self._res('decoder.mid.block_2', h)
        nres = len(dd['ch_mult'])
        for i_level in reversed(range(nres)):
            for i_block in range(dd['num_res_blocks'] + 1):
                h = self._res(f'decoder.up.{i_level}.block.{i_block}', h)
                if f'decoder.up.{i_level}.attn.{i_block}.norm.weight' in self.p:
                    h = self._attn(f'decoder.up.{i_level}.attn.{i_block}', h)
            if i_level != 0:
                h = h.repeat(2, axis=1).repeat(2, axis=2)            # nearest x2 (layers.py:32)
                if dd.get('resamp_with_conv', True):
                    h = self._conv(f'decoder.up.{i_level}.upsample.conv', h)
        h = silu(self._norm('decoder.norm_out', h))
        return np.transpose(self._conv('decoder.conv_out', h), (0, 3, 1, 2))

    # ------------------------------------------------------------------ model API
    def encode(self, x_nchw):
        """rqvae.py:80-83 -> (B,h,w,embed_dim) NHWC"""
        return self._conv('quant_conv', self.encoder(x_nchw))

    def decode(self, z_q_nhwc):
        """rqvae.py:85-89 -> (B,3,H,W) NCHW"""
        return self.decoder(self._conv('post_quant_conv', np.asarray(z_q_nhwc, np.float32)))

    def quantize(self, z_e):
        return rq_quantize(z_e, self.codebooks)

    def get_codes(self, x_nchw):
        """rqvae.py:91-95 (to_code_shape is the identity when latent hw == code hw)"""
        return self.quantize(self.encode(x_nchw))[1]

    def embed_code(self, codes):
        return rq_embed_code(codes, self.codebooks)

    def decode_code(self, codes):
        """rqvae.py:105-109"""
        return self.decode(self.embed_code(codes))

    def forward(self, x_nchw):
        """rqvae.py:74-78 + quantizations.py:273-295 (straight-through z_q, commitment loss)."""
        z_e = self.encode(x_nchw)
        quant_list, codes = self.quantize(z_e)
        loss = np.mean([np.mean((z_e - q) ** 2, dtype=np.float32) for q in quant_list], dtype=np.float32)
        z_q = z_e + (quant_list[-1] - z_e)
        return self.decode(z_q), loss, codes
