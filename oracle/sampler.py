"""Sampler oracle (numpy).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates rqvae/utils/utils.py of the reference:
  top_k_logits        :60-64   keep logits >= k-th largest (strict '<' compare: ties survive)
  top_p_probs         :67-79   sort desc, cumsum, drop entries whose *preceding* cumulative
                               mass >= p (first crossing token kept), renormalise
  sample_from_logits  :82-123  fp32 cast, /temperature, top-k, NaN->-inf, softmax, top-p
The multinomial draw itself (utils.py:112) is RNG-stream dependent and is not part of the
oracle; parity is defined on the filtered probability vector.
"""
import numpy as np


def top_k_logits(logits, k):
    """utils.py:60-64"""
    logits = np.asarray(logits, np.float32)
    kth = np.sort(logits, axis=-1)[:, ::-1][:, k - 1:k]      # v[:, [-1]] of topk
    out = logits.copy()
    out[out < kth] = -np.inf
    return out


def softmax(x):
    x = np.asarray(x, np.float32)
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m, dtype=np.float32)
    return (e / e.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def top_p_probs(probs, p):
    """utils.py:67-79.  Tie order inside torch.sort is implementation-defined; a stable
    descending sort (lowest index first among equals) is used here."""
    probs = np.asarray(probs, np.float32)
    order = np.argsort(-probs, axis=-1, kind='stable')
    sp = np.take_along_axis(probs, order, -1)
    cum = np.cumsum(sp, axis=-1, dtype=np.float32)
    remove_sorted = cum >= np.float32(p)
    remove_sorted[:, 1:] = remove_sorted[:, :-1].copy()
    remove_sorted[:, 0] = False
    remove = np.zeros_like(remove_sorted)
    np.put_along_axis(remove, order, remove_sorted, -1)
    out = np.where(remove, np.float32(0), probs)
    return (out / out.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def filtered_probs(logits, temperature=1.0, top_k=None, top_p=None):
    """utils.py:96-110: everything of sample_from_logits up to the multinomial draw."""
    x = np.asarray(logits, np.float32) / np.float32(temperature)
    if top_k is not None:
        x = top_k_logits(x, top_k)
    x = np.where(np.isnan(x), -np.inf, x).astype(np.float32)
    probs = softmax(x)
    if top_p is not None:
        probs = top_p_probs(probs, top_p)
    return probs
