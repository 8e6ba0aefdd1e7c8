"""Residual quantiser: API mirror of rqvae/models/rqvae/quantizations.py of the reference
(VQEmbedding :24-146, RQBottleneck :149-334, inference paths).  The nearest-codebook search, residual
update and code->embedding lookups run in librqamd (csrc/quantize.hip); the EMA codebook update,
dead-code restart (:80-129) are stage-1 *training* code and out of scope -- they raise
NotImplementedError; get_soft_codes (:371-400) runs natively."""
from typing import Iterable

import numpy as np
import torch
from torch import nn

from ... import _native


class VQEmbedding(nn.Embedding):
    """quantizations.py:24-41: (n_embed + 1, embed_dim) table whose last row is the zero padding index."""

    def __init__(self, n_embed, embed_dim, ema=True, decay=0.99, restart_unused_codes=True, eps=1e-5):
        super().__init__(n_embed + 1, embed_dim, padding_idx=n_embed)
        self.ema, self.decay, self.eps = ema, decay, eps
        self.restart_unused_codes = restart_unused_codes
        self.n_embed = n_embed
        if self.ema:
            _ = [p.requires_grad_(False) for p in self.parameters()]
            self.register_buffer('cluster_size_ema', torch.zeros(n_embed))
            self.register_buffer('embed_ema', self.weight[:-1, :].detach().clone())

    def codebook(self):
        """weight[:-1] (quantizations.py:45): the searchable rows, a contiguous view."""
        return self.weight.detach()[:-1]

    def code_norms(self):
        """||c||^2 per code (the codebook term of compute_distances, quantizations.py:51-52), cached per codebook
        version: recomputed after load_state_dict / .to(device) / .float() (hooked below) and after any in-place edit that
        bumps the tensor's version counter.  Writes through ``weight.data`` (``weight.data.copy_()``, EMA helpers that edit
        ``.data``) do NOT bump it: call ``invalidate_code_norms()`` after such an edit."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        if getattr(self, '_norm_key', None) != key:
            self._norms = _native.rq_code_norms(self.codebook())
            self._norm_key = key
        return self._norms

    def invalidate_code_norms(self):
        """Forget the cached ||c||^2 (next quantize recomputes them): required after edits the version counter cannot see."""
        self._norm_key = None
        self._norms = None

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_code_norms()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_code_norms()
        return super()._apply(fn, *args, **kwargs)

    @torch.no_grad()
    def find_nearest_embedding(self, inputs):
        """quantizations.py:64-69"""
        shape = inputs.shape
        assert shape[-1] == self.weight.shape[1]
        x = inputs.detach().reshape(-1, shape[-1]).to(torch.float32).contiguous()
        codes, _ = _native.rq_quantize(x, [self.codebook()], want_quants=False, norms=[self.code_norms()])
        return codes.reshape(shape[:-1])

    @torch.no_grad()
    def forward(self, inputs):
        """quantizations.py:131-142 (eval branch)."""
        if self.training and self.ema:
            raise NotImplementedError('EMA codebook updates (training) are outside the sampling path')
        embed_idxs = self.find_nearest_embedding(inputs)
        return self.embed(embed_idxs), embed_idxs

    def embed(self, idxs):
        """quantizations.py:144-146"""
        flat = idxs.reshape(-1, 1).contiguous()
        out = _native.rq_embed(flat, [self.codebook()], 0)
        return out.reshape(*idxs.shape, self.weight.shape[1])


class RQBottleneck(nn.Module):
    """quantizations.py:149-214 constructor semantics (same ValueErrors / asserts)."""

    def __init__(self, latent_shape, code_shape, n_embed, decay=0.99, shared_codebook=False, restart_unused_codes=True,
                 commitment_loss='cumsum'):
        super().__init__()
        if not len(code_shape) == len(latent_shape) == 3:
            raise ValueError("incompatible code shape or latent shape")
        if any([y % x != 0 for x, y in zip(code_shape[:2], latent_shape[:2])]):
            raise ValueError("incompatible code shape or latent shape")
        embed_dim = int(np.prod(latent_shape[:2]) // np.prod(code_shape[:2]) * latent_shape[2])
        self.latent_shape = torch.Size(latent_shape)
        self.code_shape = torch.Size(code_shape)
        self.shape_divisor = torch.Size([latent_shape[i] // code_shape[i] for i in range(len(latent_shape))])
        self.shared_codebook = shared_codebook
        if self.shared_codebook:
            if isinstance(n_embed, Iterable) or isinstance(decay, Iterable):
                raise ValueError("Shared codebooks are incompatible with list types of momentums or sizes: Change it into int")
        self.restart_unused_codes = restart_unused_codes
        self.n_embed = n_embed if isinstance(n_embed, Iterable) else [n_embed for _ in range(self.code_shape[-1])]
        self.decay = decay if isinstance(decay, Iterable) else [decay for _ in range(self.code_shape[-1])]
        assert len(self.n_embed) == self.code_shape[-1]
        assert len(self.decay) == self.code_shape[-1]
        if self.shared_codebook:
            codebook0 = VQEmbedding(self.n_embed[0], embed_dim, decay=self.decay[0], restart_unused_codes=restart_unused_codes)
            self.codebooks = nn.ModuleList([codebook0 for _ in range(self.code_shape[-1])])
        else:
            self.codebooks = nn.ModuleList([VQEmbedding(self.n_embed[i], embed_dim, decay=self.decay[i],
                                                        restart_unused_codes=restart_unused_codes)
                                            for i in range(self.code_shape[-1])])
        self.commitment_loss = commitment_loss

    # ---- shape helpers (quantizations.py:216-235)
    def to_code_shape(self, x):
        (B, H, W, D) = x.shape
        (rH, rW, _) = self.shape_divisor
        x = x.reshape(B, H // rH, rH, W // rW, rW, D).permute(0, 1, 3, 2, 4, 5)
        return x.reshape(B, H // rH, W // rW, -1)

    def to_latent_shape(self, x):
        (B, h, w, _) = x.shape
        (_, _, D) = self.latent_shape
        (rH, rW, _) = self.shape_divisor
        x = x.reshape(B, h, w, rH, rW, D).permute(0, 1, 3, 2, 4, 5)
        return x.reshape(B, h * rH, w * rW, D)

    def codebook_list(self):
        return [cb.codebook() for cb in self.codebooks]

    def _norm_list(self):
        return [cb.code_norms() for cb in self.codebooks]

    def _no_training(self):
        # the reference's train-mode quantize() runs VQEmbedding.forward, which updates the EMA codebook statistics and
        # restarts dead codes (quantizations.py:80-129,131-142); that is stage-1 training, not this path -- refuse loudly
        # rather than quantise with codebooks that silently never update
        if self.training and any(cb.ema for cb in self.codebooks):
            raise NotImplementedError('RQBottleneck in train mode updates its EMA codebooks (stage-1 training, out of scope): '
                                      'call .eval() for the sampling / reconstruction path')

    # ---- the hot path
    def quantize(self, x):
        """quantizations.py:237-271 -> (quant_list: depth x (B,h,w,D) cumulative, codes (B,h,w,depth) int64)"""
        self._no_training()
        B, h, w, embed_dim = x.shape
        flat = x.detach().reshape(-1, embed_dim).to(torch.float32).contiguous()
        codes, quants = _native.rq_quantize(flat, self.codebook_list(), want_quants=True, norms=self._norm_list())
        quant_list = [quants[i].reshape(B, h, w, embed_dim) for i in range(quants.shape[0])]
        return quant_list, codes.reshape(B, h, w, -1)

    def get_codes_only(self, x):
        B, h, w, embed_dim = x.shape
        flat = x.detach().reshape(-1, embed_dim).to(torch.float32).contiguous()
        codes, _ = _native.rq_quantize(flat, self.codebook_list(), want_quants=False, norms=self._norm_list())
        return codes.reshape(B, h, w, -1)

    def forward(self, x):
        """quantizations.py:273-281"""
        x_reshaped = self.to_code_shape(x)
        quant_list, codes = self.quantize(x_reshaped)
        commitment_loss = self.compute_commitment_loss(x_reshaped, quant_list)
        quants_trunc = self.to_latent_shape(quant_list[-1])
        quants_trunc = x + (quants_trunc - x).detach()
        return quants_trunc, commitment_loss, codes

    def compute_commitment_loss(self, x, quant_list):
        """quantizations.py:283-295 (tiny reductions; plain torch ops on the device)."""
        loss_list = [(x - quant.detach()).pow(2.0).mean() for quant in quant_list]
        return torch.mean(torch.stack(loss_list))

    @torch.no_grad()
    def embed_code(self, code):
        """quantizations.py:297-311"""
        assert code.shape[1:] == self.code_shape
        flat = code.reshape(-1, code.shape[-1]).contiguous()
        embeds = _native.rq_embed(flat, self.codebook_list(), 0).reshape(*code.shape[:-1], -1)
        return self.to_latent_shape(embeds)

    @torch.no_grad()
    def embed_code_with_depth(self, code, to_latent_shape=False):
        """quantizations.py:313-334"""
        assert code.shape[-1] == self.code_shape[-1]
        flat = code.reshape(-1, code.shape[-1]).contiguous()
        embeds = _native.rq_embed(flat, self.codebook_list(), 1).reshape(*code.shape, -1)
        if to_latent_shape:
            embeds = torch.cat([self.to_latent_shape(e.squeeze(-2)).unsqueeze(-2) for e in embeds.chunk(code.shape[-1], -2)], -2)
        return embeds, None

    @torch.no_grad()
    def embed_partial_code(self, code, code_idx, decode_type='select'):
        """quantizations.py:336-369"""
        assert code.shape[1:] == self.code_shape
        assert code_idx < code.shape[-1]
        B, h, w, _ = code.shape
        per_depth, _ = self.embed_code_with_depth(code)
        if decode_type == 'select':
            embeds = per_depth[..., code_idx, :]
        elif decode_type == 'add':
            embeds = per_depth[..., :code_idx + 1, :].sum(-2)
        else:
            raise NotImplementedError(f"{decode_type} is not implemented in partial decoding")
        return self.to_latent_shape(embeds)

    @torch.no_grad()
    def get_soft_codes(self, x, temp=1.0, stochastic=False):
        """quantizations.py:371-400 -> (soft_code (B,h,w,depth,K) fp32, code (B,h,w,depth) int64)"""
        x = self.to_code_shape(x)
        B, h, w, embed_dim = x.shape
        flat = x.detach().reshape(-1, embed_dim).to(torch.float32).contiguous()
        seed, offset = 0, 0
        if stochastic:
            from ..rqtransformer.transformers import RQTransformer
            seed, offset = RQTransformer._draw_rng(x.device, 4 * len(self.codebooks))
        soft, codes = _native.rq_soft_codes(flat, self.codebook_list(), self._norm_list(), temp, stochastic, seed, offset)
        return soft.reshape(B, h, w, soft.shape[1], soft.shape[2]), codes.reshape(B, h, w, -1)
