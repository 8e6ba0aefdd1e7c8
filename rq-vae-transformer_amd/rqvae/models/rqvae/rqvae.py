"""Stage-1 model: API mirror of rqvae/models/rqvae/rqvae.py:26-168 of the reference.

Parameters live in nn.Modules under the reference's state_dict names (398 keys for the released
config), so reference checkpoints load with ``load_state_dict``.  encode / decode / decode_code /
get_codes / forward run in librqamd (csrc/engine_vae.hip, csrc/quantize.hip) on the current HIP
stream; results are bf16-compute approximations of the reference's fp32 (tolerances in tests/)."""
import torch
from torch import nn
from torch.nn import functional as F

from ... import _native
from .._sync import SideStream, push_all, signature
from ..interfaces import Stage1Model
from .layers import ResnetBlock
from .modules import Decoder, Encoder
from .quantizations import RQBottleneck


def _plain(cfg):
    """config mapping (OmegaConf DictConfig / dict / attr-dict) -> plain dict of python scalars and lists"""
    out = {}
    for k in cfg.keys():
        v = cfg[k]
        if isinstance(v, (str, int, float, bool, type(None))):
            out[k] = v
        elif hasattr(v, 'keys'):
            out[k] = _plain(v)
        else:
            out[k] = list(v)
    return out


class RQVAE(Stage1Model):
    def __init__(self, *, embed_dim=64, n_embed=512, decay=0.99, loss_type='mse', latent_loss_weight=0.25,
                 bottleneck_type='rq', ddconfig=None, checkpointing=False, **kwargs):
        super().__init__()
        assert loss_type in ['mse', 'l1']
        self.ddconfig = _plain(ddconfig)
        self.encoder = Encoder(**self.ddconfig)
        self.decoder = Decoder(**self.ddconfig)

        def set_checkpointing(m):
            if isinstance(m, ResnetBlock):
                m.checkpointing = checkpointing
        self.encoder.apply(set_checkpointing)
        self.decoder.apply(set_checkpointing)

        if bottleneck_type == 'rq':
            latent_shape = list(kwargs['latent_shape'])
            code_shape = list(kwargs['code_shape'])
            self.quantizer = RQBottleneck(latent_shape=latent_shape, code_shape=code_shape, n_embed=n_embed, decay=decay,
                                          shared_codebook=kwargs['shared_codebook'],
                                          restart_unused_codes=kwargs['restart_unused_codes'])
            self.code_shape = code_shape
        else:
            raise ValueError("invalid 'bottleneck_type' (must be 'rq')")
        self.embed_dim = embed_dim
        self.quant_conv = nn.Conv2d(self.ddconfig["z_channels"], embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, self.ddconfig["z_channels"], 1)
        self.loss_type = loss_type
        self.latent_loss_weight = latent_loss_weight
        self._engine = None
        self._engine_sig = None
        self._side = SideStream()

    # ------------------------------------------------------------------ engine plumbing
    def _eng(self):
        sig = signature(self)
        dev = self.quant_conv.weight.device
        if self._engine is not None and self._engine.device != dev:
            self._engine.close()                        # the module moved (model.to(other device)): rebuild there
            self._engine = None
        if self._engine is None or sig != self._engine_sig:
            if self._engine is None:
                self._engine = _native.VaeEngine(self.ddconfig, self.embed_dim, device=dev)
            push_all(self, self._engine, skip_prefixes=('quantizer.',))
            self._engine_sig = sig
        return self._engine

    # ------------------------------------------------------------------ reference API
    def forward(self, xs):
        """rqvae.py:74-78"""
        z_e = self.encode(xs)
        z_q, quant_loss, code = self.quantizer(z_e)
        out = self.decode(z_q)
        return out, quant_loss, code

    @torch.no_grad()
    def encode(self, x):
        """rqvae.py:80-83: (B,3,H,W) -> (B,h,w,embed_dim) NHWC fp32"""
        x = x.detach().to(torch.float32).contiguous()
        eng = self._eng()
        return self._side.run(x.device, lambda: eng.encode(x))

    @torch.no_grad()
    def decode(self, z_q):
        """rqvae.py:85-89: (B,h,w,embed_dim) NHWC -> (B,3,H,W) fp32"""
        z_q = z_q.detach().to(torch.float32).contiguous()
        eng = self._eng()
        return self._side.run(z_q.device, lambda: eng.decode(z_q))

    @torch.no_grad()
    def get_codes(self, xs):
        """rqvae.py:91-95"""
        z_e = self.encode(xs)
        return self.quantizer.get_codes_only(self.quantizer.to_code_shape(z_e))

    @torch.no_grad()
    def get_soft_codes(self, xs, temp=1.0, stochastic=False):
        """rqvae.py:97-103"""
        assert hasattr(self.quantizer, 'get_soft_codes')
        z_e = self.encode(xs)
        return self.quantizer.get_soft_codes(z_e, temp=temp, stochastic=stochastic)

    @torch.no_grad()
    def decode_code(self, code):
        """rqvae.py:105-109"""
        z_q = self.quantizer.embed_code(code)
        return self.decode(z_q)

    def get_recon_imgs(self, xs_real, xs_recon):
        """rqvae.py:111-117"""
        xs_real = xs_real * 0.5 + 0.5
        xs_recon = torch.clamp(xs_recon * 0.5 + 0.5, 0, 1)
        return xs_real, xs_recon

    def compute_loss(self, out, quant_loss, code, xs=None, valid=False):
        """rqvae.py:119-141 (kept for API completeness; plain torch ops)."""
        if self.loss_type == 'mse':
            loss_recon = F.mse_loss(out, xs, reduction='mean')
        elif self.loss_type == 'l1':
            loss_recon = F.l1_loss(out, xs, reduction='mean')
        else:
            raise ValueError('incompatible loss type')
        loss_latent = quant_loss
        if valid:
            loss_recon = loss_recon * xs.shape[0] * xs.shape[1]
            loss_latent = loss_latent * xs.shape[0]
        loss_total = loss_recon + self.latent_loss_weight * loss_latent
        return {'loss_total': loss_total, 'loss_recon': loss_recon, 'loss_latent': loss_latent, 'codes': [code]}

    def get_last_layer(self):
        return self.decoder.conv_out.weight

    @torch.no_grad()
    def get_code_emb_with_depth(self, code):
        """rqvae.py:146-148"""
        return self.quantizer.embed_code_with_depth(code)

    @torch.no_grad()
    def decode_partial_code(self, code, code_idx, decode_type='select'):
        """rqvae.py:150-158"""
        z_q = self.quantizer.embed_partial_code(code, code_idx, decode_type)
        return self.decode(z_q)

    @torch.no_grad()
    def forward_partial_code(self, xs, code_idx, decode_type='select'):
        """rqvae.py:160-168"""
        code = self.get_codes(xs)
        return self.decode_partial_code(code, code_idx, decode_type)
