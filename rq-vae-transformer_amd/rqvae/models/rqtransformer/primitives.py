"""rqvae/models/rqtransformer/primitives.py:25-165 of the reference: TupleEmbedding, LogitMask, BatchLinear -- the stage-2
building blocks selected when shared_tok_emb / shared_cls_emb are off (no released config does that).

Inside RQTransformer they are parameter containers like every other sub-module here (same state_dict keys and shapes:
``tok_emb.weight`` (sum V_d, E) + the ``tok_emb.offsets`` buffer, ``classifier.linear.weight`` (depth, E, V_max),
``classifier.linear.bias`` (depth, V_max)); the sampling engine gathers / multiplies from them natively
(csrc/engine_rqt.hip: tok_embed_kernel, per-depth classifier matrices, mask_logits_kernel).  Their ``forward`` methods keep
the reference's semantics with plain tensor ops for stand-alone use -- they are not on the sampling path."""
from typing import Iterable, Union

import numpy as np
import torch
from torch import nn


class TupleEmbedding(nn.Embedding):
    """primitives.py:25-75: one table holding the embeddings of several dictionaries back to back; input (*, D) indices."""

    def __init__(self, num_embeddings: Union[int, Iterable[int]], embedding_dim, **kwargs):
        if 'padding_idx' in kwargs:
            raise ValueError('padding_idx argument not supported')
        if isinstance(num_embeddings, int):
            num_embeddings = (num_embeddings,)
        self.num_embeddings_per_dict = list(num_embeddings)
        super().__init__(num_embeddings=sum(self.num_embeddings_per_dict), embedding_dim=embedding_dim, **kwargs)
        self.register_buffer('offsets', None)
        self.offsets = torch.tensor(np.cumsum([0] + self.num_embeddings_per_dict[:-1]), dtype=torch.long)
        self.reset_parameters()

    def reset_parameters(self):
        self.weight.data.normal_(mean=0.0, std=0.02)

    def forward(self, x):
        (*rem, D) = x.shape
        assert D == len(self.num_embeddings_per_dict)
        return super().forward(x + self.offsets.view(*[1 for _ in rem], D))


class LogitMask(nn.Module):
    """primitives.py:78-93: -inf beyond each depth's vocabulary for (N, depth, V) logits; a no-op when all sizes are equal."""

    def __init__(self, vocab_size: Iterable[int], value=-1e6):
        super().__init__()
        self.vocab_size = list(vocab_size)
        self.mask_cond = [self.vocab_size[0]] * len(self.vocab_size) != self.vocab_size
        self.value = value

    def forward(self, logits):
        if not self.mask_cond:
            return logits
        for idx, vocab_size in enumerate(self.vocab_size):
            logits[:, idx, vocab_size:].fill_(-float('Inf'))
        return logits


class BatchLinear(nn.Module):
    """primitives.py:96-165: y_i = x_i A_i^T + b_i for i = 1..n_vectors; weight (n_vectors, in, out)."""

    def __init__(self, n_vectors, in_features, out_features, bias=True):
        super().__init__()
        self.n_vectors, self.in_features, self.out_features = n_vectors, in_features, out_features
        self.weight = nn.Parameter(torch.empty(n_vectors, in_features, out_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(n_vectors, out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        self.weight.data.normal_(mean=0.0, std=0.02)
        if self.bias is not None:
            self.bias.data.zero_()

    def forward(self, input, indices=None):
        (*rem, n_vectors, in_ch) = input.shape
        weight, bias = self.weight, self.bias
        if indices:
            assert n_vectors == len(indices)
            weight = self.weight[indices]
            bias = self.bias[indices] if self.bias is not None else None
        output = torch.einsum('bij,ijk->bik', input.reshape(-1, n_vectors, in_ch), weight)
        if bias is not None:
            output = output + bias.unsqueeze(0)
        return output.reshape(*rem, n_vectors, -1)

    def extra_repr(self):
        return 'n_vectors={}, in_features={}, out_features={}, bias={}'.format(self.n_vectors, self.in_features, self.out_features,
                                                                              self.bias is not None)
