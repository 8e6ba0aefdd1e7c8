"""rqvae/utils/utils.py of the reference: set_seed :41-48, save_pickle :11-13, and the sampler entry
points top_k_logits :60-64 / top_p_probs :67-79 / sample_from_logits :82-123 -- the latter backed by the
on-device sampler kernel (csrc/rqt_kernels.hip), which needs no host synchronisation."""
import pickle
import random

import numpy as np
import torch

from .. import _native


def save_pickle(fname, data):
    with open(fname, 'wb') as fp:
        pickle.dump(data, fp, pickle.HIGHEST_PROTOCOL)


def set_seed(seed=None):
    if seed is None:
        seed = random.getrandbits(32)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    return seed


def _rng(device, n):
    from ..models.rqtransformer.transformers import RQTransformer
    return RQTransformer._draw_rng(device, n)


def filtered_probs(logits, temperature=1.0, top_k=None, top_p=None):
    """Everything of sample_from_logits up to the multinomial draw (utils.py:96-110)."""
    _, probs = _native.sample_logits(logits.to(torch.float32).contiguous(), temperature, top_k, top_p,
                                     want_probs=True, want_samples=False)
    return probs


def top_k_logits(logits, k):
    """utils.py:60-64 (plain torch ops; the fused sampler does not call this)."""
    v, _ = torch.topk(logits, k)
    out = logits.clone()
    out[out < v[:, [-1]]] = -float('Inf')
    return out


def top_p_probs(probs, p):
    """utils.py:67-79 semantics via the device kernel: probs are already normalised, so feeding their
    logarithm through the sampler at temperature 1 reproduces the nucleus filter."""
    return filtered_probs(torch.log(probs), 1.0, None, p)


def sample_from_logits(logits, temperature=1.0, top_k=None, top_p=None):
    """utils.py:82-123: (n_samples, logit_dim) -> (n_samples,) int64."""
    logits = logits.to(dtype=torch.float32).contiguous()
    seed, offset = _rng(logits.device, 4)
    samples, _ = _native.sample_logits(logits, temperature, top_k, top_p, seed=seed, offset=offset)
    return samples.view(-1)
