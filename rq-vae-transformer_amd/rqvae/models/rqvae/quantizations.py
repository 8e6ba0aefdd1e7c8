"""Residual quantiser: API mirror of rqvae/models/rqvae/quantizations.py of the reference
(VQEmbedding :24-146, RQBottleneck :149-400).  The nearest-codebook search, residual update, code->embedding
lookups, get_soft_codes (:371-400) and -- in train mode -- the EMA codebook update with the dead-code restart
(:80-129) run in librqamd (csrc/quantize.hip).  What stays out of scope is the rest of stage-1
training (GAN trainer, losses, optimisers: SURVEY.md §2); a codebook without EMA (learned by gradient) raises."""
from typing import Iterable

import numpy as np
import torch
import torch.distributed as dist
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
    def compute_distances(self, inputs):
        """quantizations.py:43-62: squared L2 distances of every input vector to every code, (*inputs.shape[:-1], n_embed) fp32 in the
        reference's expanded form ||x||^2 + ||c||^2 - 2 x.c -- the same kernel, summation order and values that
        find_nearest_embedding takes its argmin over (csrc/quantize.hip: rqamd_rq_distances)."""
        shape = inputs.shape
        assert shape[-1] == self.weight.shape[1]
        x = inputs.detach().reshape(-1, shape[-1]).to(torch.float32).contiguous()
        dist = _native.rq_distances(x, self.codebook(), self.code_norms())
        return dist.reshape(*shape[:-1], self.n_embed)

    @torch.no_grad()
    def find_nearest_embedding(self, inputs):
        """quantizations.py:64-69"""
        shape = inputs.shape
        assert shape[-1] == self.weight.shape[1]
        x = inputs.detach().reshape(-1, shape[-1]).to(torch.float32).contiguous()
        codes, _ = _native.rq_quantize(x, [self.codebook()], want_quants=False, norms=[self.code_norms()])
        return codes.reshape(shape[:-1])

    @torch.no_grad()
    def _restart_candidates(self, vectors):
        """quantizations.py:70-77,107-115: n_embed random input vectors (tiled with a little uniform noise when the batch
        holds fewer vectors than codes), the same on every rank.  torch.rand_like / torch.randperm consume the generator in
        the reference's order."""
        n_embed, dim = self.n_embed, vectors.shape[1]
        if vectors.shape[0] < n_embed:
            reps = (n_embed + vectors.shape[0] - 1) // vectors.shape[0]
            std = vectors.new_ones(dim) * 0.01 / np.sqrt(dim)
            vectors = vectors.repeat(reps, 1)
            vectors = vectors + torch.rand_like(vectors) * std
        picked = vectors[torch.randperm(vectors.shape[0], device=vectors.device)][:n_embed].contiguous()
        if dist.is_initialized():
            dist.broadcast(picked, 0)
        return picked

    @torch.no_grad()
    def _update_buffers(self, vectors, idxs):
        """quantizations.py:80-118: per-code counts and vector sums of this batch (one native pass instead of the reference's
        (n_embed x n_vectors) one-hot matmul), summed over the ranks, folded into the EMA statistics; unused codes restart."""
        flat = vectors.detach().reshape(-1, vectors.shape[-1]).to(torch.float32).contiguous()
        count, vsum = _native.rq_ema_accumulate(flat, idxs.reshape(-1).contiguous(), self.n_embed)
        if dist.is_initialized():
            dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
            dist.all_reduce(count, op=dist.ReduceOp.SUM)
        restart = self._restart_candidates(flat) if self.restart_unused_codes else None
        _native.rq_ema_update(self.cluster_size_ema, self.embed_ema, count, vsum, restart, self.decay)

    @torch.no_grad()
    def _update_embedding(self):
        """quantizations.py:120-129"""
        n = self.cluster_size_ema.sum().reshape(1)
        new_w = torch.empty_like(self.embed_ema)
        _native.rq_ema_normalize(self.cluster_size_ema, self.embed_ema, n, self.eps, new_w)
        self.weight[:-1].copy_(new_w)               # in place through torch: bumps the version counter every cache keys on
        self.invalidate_code_norms()

    @torch.no_grad()
    def forward(self, inputs):
        """quantizations.py:131-142: nearest codes with the current weights; in train mode the EMA statistics are updated from
        this batch, the embeddings are looked up in the weights as they were, and the weights are refreshed afterwards."""
        if self.training and not self.ema:
            raise NotImplementedError('a codebook learned by gradient (ema=False) needs the training graph; only EMA codebooks are supported')
        embed_idxs = self.find_nearest_embedding(inputs)
        if self.training:
            self._update_buffers(inputs, embed_idxs)
        embeds = self.embed(embed_idxs)
        if self.training:
            self._update_embedding()
        return embeds, embed_idxs

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

    # ---- the hot path
    def quantize(self, x):
        """quantizations.py:237-271 -> (quant_list: depth x (B,h,w,D) cumulative, codes (B,h,w,depth) int64).  eval: all depths
        in one native launch.  train: the reference's depth loop over VQEmbedding.forward, because every call updates its
        (possibly shared) EMA codebook before the next depth searches it."""
        B, h, w, embed_dim = x.shape
        if self.training:
            residual = x.detach().to(torch.float32).clone()
            aggregated = torch.zeros_like(residual)
            quant_list, code_list = [], []
            for codebook in self.codebooks:
                quant, code = codebook(residual)
                residual.sub_(quant)
                aggregated.add_(quant)
                quant_list.append(aggregated.clone())
                code_list.append(code.unsqueeze(-1))
            return quant_list, torch.cat(code_list, dim=-1)
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
