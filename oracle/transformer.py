"""RQ-Transformer oracle (numpy, fp32).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates the reference's stage-2 model for the released flag set
(shared_tok_emb / shared_cls_emb / input_emb_vqvae / head_emb_vqvae / cumsum_depth_ctx):
  rqvae/models/rqtransformer/transformers.py
      forward         :113-188  teacher-forced logits (B,H,W,D,V)
      cached_forward  :190-287  one (h,w,d) step with KV caches
      sample          :294-369  256-step loop, per-depth top-k / top-p lists
  rqvae/models/rqtransformer/attentions.py
      MultiSelfAttention.forward :60-104 (scale folded into K^T, causal mask, softmax)
      AttentionBlock             :107-145 (pre-LN, 4x MLP, GELU v1 = exact erf / v2 = x*sigmoid(1.702x))
  torch.nn.LayerNorm (eps 1e-5, biased variance), torch.nn.Linear (x W^T + b).
"""
import numpy as np
from scipy.special import erf

from . import backend
from .rq import rq_embed_code_with_depth
from .sampler import filtered_probs


def layer_norm(x, w, b, eps=1e-5):
    x = x.astype(np.float32)
    mu = x.mean(-1, keepdims=True, dtype=np.float32)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=np.float32)
    return (xc / np.sqrt(var + np.float32(eps))) * w + b


def linear(x, w, b=None):
    if backend.torch_on():
        import torch
        import torch.nn.functional as F
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        return F.linear(t(x), t(w), None if b is None else t(b)).numpy()
    y = x.astype(np.float32) @ w.T
    return y if b is None else y + b


def gelu(x, version='v1'):
    """attentions.py:25-36"""
    if backend.torch_on():
        import torch
        import torch.nn.functional as F
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        return (F.gelu(t) if version == 'v1' else t * torch.sigmoid(1.702 * t)).numpy()
    if version == 'v1':
        return (0.5 * x * (1.0 + erf(x / np.sqrt(2.0)))).astype(np.float32)
    return (x / (1.0 + np.exp(-1.702 * x))).astype(np.float32)


def _softmax(x):
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


class RQTransformerOracle:
    def __init__(self, cfg, params):
        self.cfg = cfg
        self.p = {k: np.asarray(v, np.float32) for k, v in params.items()}
        self.E = cfg['embed_dim']
        self.H, self.W, self.D = cfg['block_size']
        self.V = cfg['vocab_size']
        self.block_size_cond = max(cfg.get('block_size_cond', 0), 1)
        self.nb = cfg['body']['n_layer']
        self.nh_layers = cfg['head']['n_layer']
        self.n_head_body = cfg['body']['block']['n_head']
        self.n_head_head = cfg['head']['block']['n_head']
        # AttentionBlockConfig.gelu of each stack (configs.py:21-40, attentions.py:25-36,117-122); every released config: 'v1'
        self.gelu = {'body_transformer': cfg['body']['block'].get('gelu', 'v1'), 'head_transformer': cfg['head']['block'].get('gelu', 'v1')}
        # variant flags (transformers.py:60-99): all True in every released config
        self.input_vq = cfg.get('input_emb_vqvae', True)
        self.head_vq = cfg.get('head_emb_vqvae', True)
        self.cumsum = cfg.get('cumsum_depth_ctx', True)
        self.shared_cls = cfg.get('shared_cls_emb', True)
        self.shared_tok = cfg.get('shared_tok_emb', True)
        vs = cfg['vocab_size']
        self.vocab_sizes = [vs] * self.D if isinstance(vs, int) else list(vs)
        self.V = max(self.vocab_sizes)
        self.init_cache()

    # ------------------------------------------------------------------ blocks
    def _attn(self, pre, x, n_head, past=None, caching=False):
        """attentions.py:60-104.  x (B,T,C).  past = (k,v) each (B,nh,Tp,hs) or None."""
        p = self.p
        B, T, C = x.shape
        hs = C // n_head

        def proj(n):
            return linear(x, p[f'{pre}.attn.{n}.weight'], p.get(f'{pre}.attn.{n}.bias')) \
                .reshape(B, T, n_head, hs).transpose(0, 2, 1, 3)
        k, q, v = proj('key'), proj('query'), proj('value')
        Tp = 0
        if past is not None:
            k = np.concatenate([past[0], k], 2)
            v = np.concatenate([past[1], v], 2)
            Tp = past[0].shape[2]
        att = q @ (k.transpose(0, 1, 3, 2) * np.float32(1.0 / np.sqrt(hs)))
        mask = np.tril(np.ones((Tp + T, Tp + T), bool))[Tp:Tp + T, :Tp + T]
        att = np.where(mask[None, None], att, -np.inf)
        att = _softmax(att).astype(np.float32)
        y = (att @ v).transpose(0, 2, 1, 3).reshape(B, T, C)
        y = linear(y, p[f'{pre}.attn.proj.weight'], p.get(f'{pre}.attn.proj.bias'))
        return (y, (k, v)) if caching else y

    def _block(self, pre, x, n_head, past=None, caching=False):
        """attentions.py:126-142"""
        p = self.p
        h = layer_norm(x, p[f'{pre}.ln1.weight'], p[f'{pre}.ln1.bias'])
        if caching:
            a, present = self._attn(pre, h, n_head, past, True)
        else:
            a, present = self._attn(pre, h, n_head), None
        x = x + a
        h = layer_norm(x, p[f'{pre}.ln2.weight'], p[f'{pre}.ln2.bias'])
        h = gelu(linear(h, p[f'{pre}.mlp.0.weight'], p.get(f'{pre}.mlp.0.bias')), self.gelu[pre.split('.', 1)[0]])
        x = x + linear(h, p[f'{pre}.mlp.2.weight'], p.get(f'{pre}.mlp.2.bias'))
        return x, present

    def _stack(self, name, x, n_layer, n_head):
        for i in range(n_layer):
            x, _ = self._block(f'{name}.blocks.{i}', x, n_head)
        return x

    def _stack_cached(self, name, x, n_layer, n_head, cache):
        for i in range(n_layer):
            x, cache[i] = self._block(f'{name}.blocks.{i}', x, n_head, cache[i], True)
        return x

    def _tok(self, xs):
        """tok_emb(xs): nn.Embedding (shared) or TupleEmbedding (primitives.py:25-75: per-depth tables + offsets); (...,D) -> (...,D,E)"""
        w = self.p['tok_emb.weight']
        if self.shared_tok:
            return w[xs]
        offs = np.cumsum([0] + self.vocab_sizes[:-1])
        return w[xs + offs]

    def _classifier(self, x, depth=None):
        """classifier (transformers.py:90-99,278-285): LayerNorm -> Linear, or BatchLinear (primitives.py:96-165: one
        (E, Vmax) matrix per depth) -> LogitMask (-inf beyond each depth's vocabulary).  x (..., D, E), or (..., E) with
        `depth` given (cached step)."""
        p = self.p
        h = layer_norm(x, p['classifier.layer_norm.weight'], p['classifier.layer_norm.bias'])
        if self.shared_cls:
            return linear(h, p['classifier.linear.weight'], p['classifier.linear.bias'])
        W, b = p['classifier.linear.weight'], p['classifier.linear.bias']          # (D, E, Vmax), (D, Vmax)
        # LogitMask (primitives.py:78-93) indexes `logits[:, idx, vocab_size:]`, which on forward()'s (B,H,W,D,V) tensor slices
        # the W axis beyond its end -- a no-op: the reference's teacher-forced logits are NOT masked (all Vmax columns stay
        # finite).  In cached_forward the same indexing raises for depth > 0 when the sizes differ, so masked sampling is
        # undefined upstream; this path masks columns >= vocab_size[depth] when sampling (see engine_rqt.hip).
        if depth is None:
            return np.einsum('...ie,iek->...ik', h.astype(np.float32), W) + b
        return h.astype(np.float32) @ W[depth] + b[depth]

    # ------------------------------------------------------------------ forward
    def forward(self, xs, codebooks, cond=None, return_cond_logits=False):
        """transformers.py:113-188.  return_cond_logits: for cond_len > 1 return (seq_logits, cond_logits) as the
        reference does (:150-153,:185-186): cond_classifier over the body outputs of the first cond_len-1 positions."""
        p = self.p
        xs = np.asarray(xs)
        B, H, W, D = xs.shape
        xs = xs.reshape(B, H * W, D)
        if cond is None:
            cond = np.zeros((B, self.block_size_cond), np.int64)
        cond = np.asarray(cond).reshape(B, self.block_size_cond)
        seq_len, cond_len = xs.shape[1], cond.shape[1]
        emb = rq_embed_code_with_depth(xs, codebooks) if (self.input_vq or self.head_vq) else None   # (B,T,D,Din)
        xs_emb = linear(emb, p['input_mlp.weight'], p['input_mlp.bias']) if self.input_vq else self._tok(xs)  # :139-143
        conds_emb = p['cond_emb.weight'][cond] + p['pos_emb_cond'][:, :cond_len]
        xs_emb = xs_emb.sum(-2) + p['pos_emb_hw'][:, :seq_len]
        latents = np.concatenate([conds_emb, xs_emb[:, :-1]], 1)
        latents = self._stack('body_transformer', latents, self.nb, self.n_head_body)
        spatial_ctx = latents[:, cond_len - 1:]
        if self.head_vq:
            depth_ctx = linear(np.cumsum(emb, -2, dtype=np.float32) if self.cumsum else emb,
                               p['head_mlp.weight'], p['head_mlp.bias'])      # :157-162
        else:
            depth_ctx = self._tok(xs)                                         # :163-164
        full = np.concatenate([spatial_ctx.reshape(B, seq_len, 1, -1), depth_ctx[:, :, :-1]], -2)
        full = full.reshape(B * seq_len, D, -1) + p['pos_emb_d'][:, :D]
        out = self._stack('head_transformer', full, self.nh_layers, self.n_head_head)
        seq_logits = self._classifier(out.reshape(B, H, W, D, -1))
        if return_cond_logits and cond_len > 1:
            h = layer_norm(latents[:, :cond_len - 1], p['cond_classifier.layer_norm.weight'], p['cond_classifier.layer_norm.bias'])
            return seq_logits, linear(h, p['cond_classifier.linear.weight'], p['cond_classifier.linear.bias'])
        return seq_logits

    # ------------------------------------------------------------------ cached path
    def init_cache(self):
        """transformers.py:289-292"""
        self._spatial_ctx = None
        self._body_cache = [None] * self.cfg['body']['n_layer']
        self._head_cache = [None] * self.cfg['head']['n_layer']

    def cached_forward(self, xs, codebooks, cond, sample_loc):
        """transformers.py:190-287.  xs (B,h+1,W,D) prefix; returns logits (B,V).
        The reference recomputes the embeddings of the whole prefix every step
        (:218-225,:249-257); only the rows it then uses are computed here -- same values."""
        p = self.p
        h, w, d = sample_loc
        B, _, W, D = xs.shape
        idx = h * W + w
        xs = np.asarray(xs).reshape(B, -1, D)[:, :idx + 1]
        if cond is None:
            cond = np.zeros((B, self.block_size_cond), np.int64)
        cond = np.asarray(cond).reshape(B, self.block_size_cond)
        cond_len = cond.shape[1]
        if d == 0:
            if self._spatial_ctx is None:
                emb = rq_embed_code_with_depth(xs, codebooks)
                xs_emb = linear(emb, p['input_mlp.weight'], p['input_mlp.bias']).sum(-2) \
                    + p['pos_emb_hw'][:, :idx + 1]
                conds_emb = p['cond_emb.weight'][cond] + p['pos_emb_cond'][:, :cond_len]
                latents = np.concatenate([conds_emb, xs_emb[:, :-1]], 1)[:, :cond_len + idx]
                out = self._stack_cached('body_transformer', latents, self.nb, self.n_head_body,
                                         self._body_cache)
                self._spatial_ctx = out[:, -1:]
            else:
                emb = rq_embed_code_with_depth(xs[:, idx - 1:idx], codebooks)   # position idx-1
                tok = linear(emb, p['input_mlp.weight'], p['input_mlp.bias']).sum(-2) \
                    + p['pos_emb_hw'][:, idx - 1:idx]
                self._spatial_ctx = self._stack_cached('body_transformer', tok, self.nb,
                                                       self.n_head_body, self._body_cache)
            self._head_cache = [None] * self.nh_layers
        emb_hw = rq_embed_code_with_depth(xs[:, idx], codebooks)                 # (B,D,Din)
        depth_ctx = linear(np.cumsum(emb_hw, -2, dtype=np.float32),
                           p['head_mlp.weight'], p['head_mlp.bias'])
        full = np.concatenate([self._spatial_ctx.reshape(B, 1, -1), depth_ctx[:, :-1]], -2) \
            + p['pos_emb_d'][:, :D]
        out = self._stack_cached('head_transformer', full[:, d:d + 1], self.nh_layers,
                                 self.n_head_head, self._head_cache)
        return self._classifier(out).reshape(B, -1)

    def sample(self, partial_sample, codebooks, cond=None, start_loc=(0, 0), temperature=1.0,
               top_k=None, top_p=None, rng=None, max_steps=None, return_probs=False):
        """transformers.py:294-369.  Draws use numpy's Generator (the reference uses
        torch.multinomial on the global generator -- RNG streams are not comparable, parity
        is on filtered probabilities).  ``max_steps`` bounds the loop for timing samples."""
        rng = rng or np.random.default_rng(0)
        H, W, D = self.H, self.W, self.D
        xs = np.array(partial_sample, dtype=np.int64)
        assert xs.shape[1:] == (H, W, D)
        tk = [self.V] * D if top_k is None else [min(int(top_k), self.V)] * D
        tp = [1.0] * D if top_p is None else [min(float(top_p), 1.0)] * D
        self.init_cache()
        steps, probs_log = 0, []
        for h in range(H):
            for w in range(W):
                for d in range(D):
                    if (h, w) < tuple(start_loc):
                        continue
                    if max_steps is not None and steps >= max_steps:
                        self.init_cache()
                        return (xs, probs_log) if return_probs else xs
                    logits = self.cached_forward(xs[:, :h + 1], codebooks, cond, (h, w, d))
                    probs = filtered_probs(logits, temperature, tk[d], tp[d])
                    if return_probs:
                        probs_log.append(probs)
                    cdf = np.cumsum(probs.astype(np.float64), -1)
                    u = rng.random((xs.shape[0], 1)) * cdf[:, -1:]
                    xs[:, h, w, d] = np.minimum((cdf < u).sum(-1), self.V - 1)
                    steps += 1
        self.init_cache()
        return (xs, probs_log) if return_probs else xs
