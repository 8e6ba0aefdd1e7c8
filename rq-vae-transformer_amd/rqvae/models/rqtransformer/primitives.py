"""Stage-2 building blocks that the reference selects with shared_tok_emb / shared_cls_emb = False
(rqvae/models/rqtransformer/primitives.py:25-165 there: TupleEmbedding, LogitMask, BatchLinear; no released config does).

Here they are what every other sub-module of RQTransformer is: holders of parameters under the reference's state_dict names and
shapes -- ``tok_emb.weight`` (sum of the per-depth vocabularies, E) with the integer buffer ``tok_emb.offsets``,
``classifier.linear.weight`` (depth, E, V_max), ``classifier.linear.bias`` (depth, V_max) -- so that reference checkpoints load
with ``strict=True``.  The arithmetic lives in the engine (csrc/engine_rqt.hip: ``tok_embed_kernel`` gathers and sums the per-depth
rows, the classifier GEMM runs on the matrix of the current depth, ``mask_logits_kernel`` is the LogitMask of the sampling path).
Stand-alone the three modules still compute what their reference namesakes compute -- on any device and under autograd -- as a few
torch calls (the lookup takes the library's gather, ``rqamd_rq_embed``, for fp32 CUDA inference): ``model.tok_emb(x)``,
``model.classifier.linear(h)`` and ``model.classifier.logit_mask(logits)`` keep working for code that pokes at sub-modules, none of
it on the sampling path."""
import itertools

import torch
from torch import nn

from ... import _native


class TupleEmbedding(nn.Embedding):
    """One table for several dictionaries laid end to end (reference primitives.py:25-75).  ``forward`` takes (..., D) indices, one
    per dictionary, and returns (..., D, E) rows -- gathered by the library's embedding kernel from per-dictionary views of the table."""

    def __init__(self, num_embeddings, embedding_dim, **kwargs):
        if 'padding_idx' in kwargs:
            raise ValueError('padding_idx argument not supported')
        sizes = [num_embeddings] if isinstance(num_embeddings, int) else [int(n) for n in num_embeddings]
        self.num_embeddings_per_dict = sizes
        super().__init__(num_embeddings=sum(sizes), embedding_dim=embedding_dim, **kwargs)
        starts = [0] + list(itertools.accumulate(sizes))[:-1]
        self.register_buffer('offsets', torch.tensor(starts, dtype=torch.long))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.weight, mean=0.0, std=0.02)

    def forward(self, x):
        sizes = self.num_embeddings_per_dict
        assert x.shape[-1] == len(sizes)
        w = self.weight
        native = (w.is_cuda and x.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
                  and not (torch.is_grad_enabled() and w.requires_grad))
        if not native:
            # any device / under autograd: row (code + start of its dictionary) of the one table
            return nn.functional.embedding(x.to(torch.long) + self.offsets.to(x.device), w)
        tables = [w.detach()[o:o + n] for o, n in zip(self.offsets.tolist(), sizes)]
        flat = x.reshape(-1, len(sizes)).to(torch.long).contiguous()
        out = _native.rq_embed(flat, tables, 1)                     # mode 1: one row per depth, not summed
        return out.reshape(*x.shape, self.embedding_dim)


class LogitMask(nn.Module):
    """Which depths have a vocabulary smaller than the widest one (reference primitives.py:78-93); the engine masks the columns
    beyond ``vocab_size[d]`` when it samples depth d."""

    def __init__(self, vocab_size, value=-1e6):
        super().__init__()
        self.vocab_size = [int(v) for v in vocab_size]
        self.mask_cond = len(set(self.vocab_size)) > 1
        self.value = value

    def forward(self, logits):
        """(B, depth, V) logits, in place: depth d keeps its first vocab_size[d] columns, the rest become -inf"""
        if self.mask_cond:
            for d, v in enumerate(self.vocab_size):
                logits[:, d, v:] = float('-inf')
        return logits


class BatchLinear(nn.Module):
    """Per-depth classifier matrices (reference primitives.py:96-165): ``weight`` (n_vectors, in_features, out_features),
    ``bias`` (n_vectors, out_features) or None."""

    def __init__(self, n_vectors, in_features, out_features, bias=True):
        super().__init__()
        self.n_vectors, self.in_features, self.out_features = n_vectors, in_features, out_features
        self.weight = nn.Parameter(torch.empty(n_vectors, in_features, out_features))
        self.bias = nn.Parameter(torch.empty(n_vectors, out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.normal_(self.weight, mean=0.0, std=0.02)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, input, indices=None):
        """(..., n, in_features) -> (..., n, out_features): vector i through matrix i (through matrix indices[i] when given)"""
        w = self.weight if not indices else self.weight[list(indices)]
        b = None if self.bias is None else (self.bias if not indices else self.bias[list(indices)])
        lead, n = input.shape[:-2], input.shape[-2]
        assert n == w.shape[0]
        rows = input.reshape(-1, n, input.shape[-1]).transpose(0, 1)          # (n, batch, in)
        out = torch.bmm(rows, w).transpose(0, 1)                              # (batch, n, out)
        if b is not None:
            out = out + b
        return out.reshape(*lead, n, w.shape[-1])

    def extra_repr(self):
        return f'n_vectors={self.n_vectors}, in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}'
