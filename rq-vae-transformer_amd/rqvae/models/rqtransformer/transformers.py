"""Stage-2 model: API mirror of rqvae/models/rqtransformer/transformers.py:34-369 of the reference.

``sample`` keeps the reference signature (transformers.py:294-308) and semantics -- partial_sample is
cloned not mutated, rows before start_loc are kept, per-depth top-k/top-p lists, cond None -> zeros --
but the 256-step loop runs inside librqamd's sampling engine (csrc/engine_rqt.hip): one C call enqueues
the whole loop on a side stream, replaying one captured hipGraph per spatial position, with the
sampler on the device (no host sync per step).  Weights, GEMM operands and the KV cache are bf16 by default
(BASELINE.json's compute dtype) and IEEE fp16 under ``amp=True`` -- the reference's ``amp`` is fp16 autocast
(transformers.py:21,206; main_sampling_fid.py:216 passes it) and the engine's fp16 build (librqamd_f16.so) is what
serves it: three more mantissa bits than bf16, the same MFMA rate; accumulation / residual stream / LayerNorm /
softmax / logits are fp32 either way.  Arguments of ``sample`` that cannot mean here what they mean upstream
are never ignored silently: ``cached=False`` runs a real uncached loop (every step recomputes all logits; the reference's own
cross-check of its cache, transformers.py:352-356); ``is_tqdm`` / ``desc`` have nothing to report on (the
whole loop is one asynchronous C call) and ``fast`` is unused by the reference itself."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _native
from .._sync import SideStream, push_all, signature
from ..interfaces import Stage2Model
from .attentions import AttentionStack
from .configs import resolve
from .primitives import BatchLinear, LogitMask, TupleEmbedding


class _AttrView(dict):
    """resolved config with attribute access (the reference keeps an OmegaConf node in self.config)"""
    __getattr__ = dict.__getitem__

    def copy(self):
        return _attr(dict(self))


def _attr(d):
    return _AttrView({k: _attr(v) if isinstance(v, dict) else v for k, v in d.items()})


class RQTransformer(Stage2Model):

    def __init__(self, config):
        super().__init__()
        cfg = resolve(config)
        if len(cfg['block_size']) != 3:
            raise ValueError("incompatible block size")
        self.block_size = torch.Size(cfg['block_size'])
        if isinstance(cfg['vocab_size'], int):
            cfg['vocab_size'] = [cfg['vocab_size']] * cfg['block_size'][2]
        if cfg['shared_tok_emb'] or cfg['shared_cls_emb']:
            assert [cfg['vocab_size'][0]] * len(cfg['vocab_size']) == list(cfg['vocab_size'])
        self.config = _attr(cfg)
        self.vocab_size = list(cfg['vocab_size'])
        # every released config sets input_emb_vqvae / head_emb_vqvae / shared_cls_emb / cumsum_depth_ctx
        # (configs/**/stage2/*.yaml:13-18); the other combinations (primitives.py) are supported by the engine as well
        E = cfg['embed_dim']
        self.vocab_size_cond = max(cfg['vocab_size_cond'], 1)
        self.block_size_cond = max(cfg['block_size_cond'], 1)
        assert not (self.block_size_cond > 1 and self.vocab_size_cond == 1)
        self.cond_emb = nn.Embedding(self.vocab_size_cond, E)
        self.tok_emb, self.input_mlp, self.head_mlp = None, None, None          # transformers.py:60-70
        if cfg['input_emb_vqvae']:
            self.input_mlp = nn.Linear(cfg['input_embed_dim'], E)
        if cfg['head_emb_vqvae']:
            self.head_mlp = nn.Linear(cfg['input_embed_dim'], E)
        if not (cfg['input_emb_vqvae'] and cfg['head_emb_vqvae']):
            if cfg['shared_tok_emb']:
                self.tok_emb = nn.Embedding(cfg['vocab_size'][0], E)
            else:
                self.tok_emb = TupleEmbedding(cfg['vocab_size'], E)
        self.pos_emb_cond = nn.Parameter(torch.zeros(1, self.block_size_cond, E))
        self.pos_emb_hw = nn.Parameter(torch.zeros(1, self.block_size[0] * self.block_size[1], E))
        self.pos_emb_d = nn.Parameter(torch.zeros(1, self.block_size[2], E))
        self.pos_emb_cond.data.normal_(mean=0.0, std=0.02)
        self.pos_emb_hw.data.normal_(mean=0.0, std=0.02)
        self.pos_emb_d.data.normal_(mean=0.0, std=0.02)
        self.embed_drop = nn.Dropout(cfg['embd_pdrop'], inplace=True)
        self.body_transformer = AttentionStack(cfg['body'])
        self.head_transformer = AttentionStack(cfg['head'])
        self.classifier = nn.Sequential(OrderedDict([
            ('layer_norm', nn.LayerNorm(E)),
            ('linear', nn.Linear(E, cfg['vocab_size'][0]) if cfg['shared_cls_emb']
             else BatchLinear(cfg['block_size'][2], E, max(cfg['vocab_size']))),
            ('logit_mask', LogitMask(cfg['vocab_size'], value=-1e6)),
        ]))
        if cfg['block_size_cond'] > 1:
            self.cond_classifier = nn.Sequential(OrderedDict([
                ('layer_norm', nn.LayerNorm(E)),
                ('linear', nn.Linear(E, cfg['vocab_size_cond'])),
            ]))
        self._cache = None
        self._engines = {}                               # amp (False: bf16 engine, True: fp16 engine) -> (engine, parameter signature)
        self._side = SideStream()
        self.use_graph = os.environ.get('RQAMD_GRAPH', '1') != '0'
        # 'philox' (default): on-device sampler inside the captured graphs; 'torch': host loop + torch.multinomial (see sample)
        self.sampler = os.environ.get('RQAMD_SAMPLER', 'philox')

    # ------------------------------------------------------------------ engine plumbing
    @property
    def _engine(self):
        """the default (bf16) engine, or None before its first use (bench.py / scripts read its profile counters)"""
        return self._engines.get(False, (None, None))[0]

    @_engine.setter
    def _engine(self, value):
        if value is None:
            self._engines.pop(False, None)
        else:
            self._engines[False] = (value, None)

    def _eng(self, amp=False):
        """The native engine for this precision: amp=False -> bf16 (librqamd.so), amp=True -> IEEE fp16 (librqamd_f16.so).  Each keeps its
        own packed copy of the parameters (pushed again whenever the module's tensors change) and its own KV cache / graphs."""
        amp = bool(amp)
        sig = signature(self)
        eng, esig = self._engines.get(amp, (None, None))
        if eng is not None and eng.device != self.pos_emb_hw.device:
            eng.close()                                 # the module moved (model.to(other device)): rebuild there
            eng = None
        if eng is None or sig != esig:
            self._cf = None                             # (a cached_forward sequence does not survive an engine rebuild / parameter push)
            if eng is None:
                c = self.config
                gb, gh = c.body.block.gelu == 'v2', c.head.block.gelu == 'v2'
                gelu_code = 1 if (gb and gh) else 3 if gb else 2 if gh else 0      # (the stacks may differ: include/rqamd.h)
                eng = _native.RqtEngine(
                    embed_dim=c.embed_dim, n_head=c.body.block.n_head, n_layer_body=c.body.n_layer, n_layer_head=c.head.n_layer,
                    vocab_size=max(self.vocab_size), input_embed_dim=c.input_embed_dim, vocab_size_cond=self.vocab_size_cond,
                    block_size_cond=self.block_size_cond, block_size=list(self.block_size), gelu_v2=gelu_code,
                    device=self.pos_emb_hw.device, input_emb_vqvae=c.input_emb_vqvae, head_emb_vqvae=c.head_emb_vqvae,
                    shared_tok_emb=c.shared_tok_emb, shared_cls_emb=c.shared_cls_emb, cumsum_depth_ctx=c.cumsum_depth_ctx,
                    vocab_sizes=self.vocab_size, half=amp, n_head_head=c.head.block.n_head)
            push_all(self, eng)
            # bias-free layers (attn_bias / mlp_bias = False): the engine's epilogues always add a bias vector -- zeros here
            dev = self.pos_emb_hw.device
            for prefix, stack in (('body_transformer', self.body_transformer), ('head_transformer', self.head_transformer)):
                for i, blk in enumerate(stack.blocks):
                    for leaf, lin in (('attn.query', blk.attn.query), ('attn.key', blk.attn.key), ('attn.value', blk.attn.value),
                                      ('attn.proj', blk.attn.proj), ('mlp.0', blk.mlp[0]), ('mlp.2', blk.mlp[2])):
                        if lin.bias is None:
                            eng.set_param(f'{prefix}.blocks.{i}.{leaf}.bias', torch.zeros(lin.out_features, device=dev))
            self._engines[amp] = (eng, sig)
        return eng

    @staticmethod
    def _codebooks(model_aux):
        """model_aux.get_code_emb_with_depth is the only thing the reference uses model_aux for
        (transformers.py:109-111); the engine gathers from the same tables directly."""
        if model_aux is None or not hasattr(model_aux, 'quantizer'):
            raise ValueError('model_aux must be the RQVAE whose codebook embeds the codes (input_emb_vqvae=True)')
        return model_aux.quantizer.codebook_list()

    def _checked_codebooks(self, model_aux):
        """the engine gathers rows `code` < vocab_size of width input_embed_dim from these tables: anything else would read
        out of bounds, so the shapes are checked here (the reference would fail in F.embedding / input_mlp)"""
        D = self.block_size[2]
        if not (self.config.input_emb_vqvae or self.config.head_emb_vqvae):
            return [self.tok_emb.weight.detach()] * D      # learned embeddings only: model_aux is not consulted (placeholders)
        cbs = self._codebooks(model_aux)
        if len(cbs) < D:
            raise ValueError(f'model_aux has {len(cbs)} codebooks, the transformer needs {D}')
        for d in range(D):
            cb = cbs[d]
            if cb.dim() != 2 or cb.shape[1] != self.config.input_embed_dim or cb.shape[0] < self.vocab_size[d]:
                raise ValueError(f'codebook {d} has shape {tuple(cb.shape)}; expected (>= {self.vocab_size[d]}, '
                                 f'{self.config.input_embed_dim})')
        return cbs[:D]

    def _cond(self, cond, B, device):
        if cond is None:
            return None                                  # engine zero-fills (transformers.py:208-209)
        return cond.reshape(B, self.block_size_cond).to(device=device, dtype=torch.long).contiguous()

    def _on_side_stream(self, device, fn):
        return self._side.run(device, fn)

    # ------------------------------------------------------------------ reference API
    def init_cache(self):
        """transformers.py:289-292 -- caches are engine-side and reset by every sample()/forward(); a cached_forward sequence in
        progress ends here."""
        self._cache = {'spatial_ctx_hw': None}
        self._cf = None

    @torch.no_grad()
    def cached_forward(self, xs, model_aux=None, cond=None, amp=False, sample_loc=(0, 0, 0)):
        """transformers.py:190-287: logits (B, vocab_size) fp32 of ONE step (h, w, d) from the KV caches of the steps before it, over
        the engine's stepping entry points (rqamd_rqt_step_begin / _step_logits / _step_set_code: the arithmetic of sample()).  As in
        the reference the calls of a sequence come in sampling order after init_cache(): a call with d == 0 that does not continue
        the previous call starts a sequence -- the codes of the positions before (h, w) are taken from `xs` and only feed the body
        stack's KV cache (the start_loc > (0, 0) prefill, :235-239; that restart costs `pos` body-only steps, and so does any call that
        finds the engine rebuilt, the parameters pushed again or `amp` changed since the previous one) -- and a call with d > 0 must
        follow (h, w, d - 1).  `model_aux` and `cond` are read when a sequence starts; only the code of the step before is re-read
        from `xs` at a continuing call (the reference re-embeds all of `xs` every time).  `xs` holds
        the codes drawn so far, (B, h + 1 .. H, W, D) as sample() passes them (:349); the codes of the step before are read from it
        at every call, so a caller may write them in place like sample() does (:364).  The returned tensor is the caller's own."""
        (h, w, d) = (int(v) for v in sample_loc)
        (B, Hx, W, D) = xs.shape
        (H, W_, D_) = self.block_size
        assert (W, D) == (W_, D_) and h < Hx <= H and 0 <= w < W and 0 <= d < D
        pos = h * W + w
        eng = self._eng(amp)
        st = getattr(self, '_cf', None)
        if st is not None and st.get('eng') is not eng:
            st = None                                   # another engine (precision) or a rebuilt one: the sequence restarts
        if st is None or st['next'] != (pos, d) or st['B'] != B:
            if d != 0:
                raise RuntimeError(f'cached_forward(sample_loc={tuple(sample_loc)}): depth {d} must follow depth {d - 1} of the same position '
                                   '(the head stack attends its cached depths; call init_cache() and step in sampling order)')
            cbs = self._checked_codebooks(model_aux)
            full = torch.zeros((B, H, W, D), dtype=torch.long, device=xs.device)
            full[:, :Hx] = xs
            c = self._cond(cond, B, xs.device)
            if c is None:
                c = torch.zeros((B, self.block_size_cond), dtype=torch.long, device=xs.device)
            eng.step_begin(full.contiguous(), c, cbs)
            for p in range(pos):
                eng.step_logits(p, -1)                 # given codes: body KV cache only
        elif d > 0:
            eng.step_set_code(pos, d - 1, xs[:, h, w, d - 1].to(torch.long).contiguous())
        elif pos > 0:
            hp, wp = divmod(pos - 1, W)
            eng.step_set_code(pos - 1, D - 1, xs[:, hp, wp, D - 1].to(torch.long).contiguous())
        logits = eng.step_logits(pos, d).clone()
        self._cf = {'next': (pos, d + 1) if d + 1 < D else (pos + 1, 0), 'B': B, 'eng': eng}
        return logits

    def embed_with_model_aux(self, xs, model_aux):
        xs_emb, _ = model_aux.get_code_emb_with_depth(xs)
        return xs_emb

    @torch.no_grad()
    def forward(self, xs, model_aux=None, cond=None, amp=False):
        """transformers.py:113-188: teacher-forced logits (B,H,W,D,V) fp32, computed by stepping the
        engine's cached path over the given codes (identical to the uncached pass up to rounding --
        the reference's own cached==uncached invariant, transformers.py:352-356)."""
        self._cf = None                                  # (any other engine call ends a cached_forward sequence)
        if self.block_size_cond > 1:
            # (seq_logits, cond_logits): cond_classifier over the body outputs of the first cond_len-1 positions
            # (transformers.py:150-153,185-186); the engine takes them from the multi-token prefill of the prefix
            (B, H, W, D) = xs.shape
            assert torch.Size([H, W, D]) == self.block_size
            eng = self._eng(amp)
            cbs = self._checked_codebooks(model_aux)
            codes = xs.to(torch.long).contiguous()
            c = self._cond(cond, B, xs.device)
            if c is None:
                c = torch.zeros((B, self.block_size_cond), dtype=torch.long, device=xs.device)
            return self._on_side_stream(xs.device, lambda: eng.forward(codes, c, cbs))
        return self.teacher_forced_logits(xs, model_aux, cond, amp=amp)

    @torch.no_grad()
    def teacher_forced_logits(self, xs, model_aux=None, cond=None, amp=False):
        """seq_logits of forward() (transformers.py:113-188) for any block_size_cond, via the engine's cached path."""
        (B, H, W, D) = xs.shape
        assert torch.Size([H, W, D]) == self.block_size
        self._cf = None
        eng = self._eng(amp)
        cbs = self._checked_codebooks(model_aux)
        codes = xs.to(torch.long).contiguous()
        c = self._cond(cond, B, xs.device)
        return self._on_side_stream(xs.device, lambda: eng.logits(codes, c, cbs))

    @torch.no_grad()
    def sample(self, partial_sample, model_aux=None, cond=None, start_loc=(0, 0), temperature=1.0, top_k=None, top_p=None,
               amp=False, cached=True, is_tqdm=False, desc="Sampling", fast=True):
        """transformers.py:294-369"""
        assert self.block_size == partial_sample.shape[1:]
        self._cf = None
        (H, W, D) = self.block_size
        if top_k is None:
            top_k_list = [self.vocab_size[i] for i in range(D)]
        elif isinstance(top_k, int):
            top_k_list = [min(top_k, self.vocab_size[i]) for i in range(D)]
        elif len(top_k) == 1:
            top_k_list = [min(top_k[0], self.vocab_size[i]) for i in range(D)]
        else:
            top_k_list = [min(top_k[i], self.vocab_size[i]) for i in range(D)]
        if top_p is None:
            top_p_list = [1.0 for _ in range(D)]
        elif isinstance(top_p, float):
            top_p_list = [min(top_p, 1.0) for _ in range(D)]
        elif len(top_p) == 1:
            top_p_list = [min(top_p[0], 1.0) for _ in range(D)]
        else:
            top_p_list = [min(top_p[i], 1.0) for i in range(D)]
        B = partial_sample.shape[0]
        device = partial_sample.device
        eng = self._eng(amp)                            # amp=True: the fp16 build of the engine (the reference's fp16 autocast)
        cbs = self._checked_codebooks(model_aux)
        xs = partial_sample.to(torch.long).contiguous()
        c = self._cond(cond, B, device)
        if self.sampler == 'torch':
            if not cached:
                raise NotImplementedError("sampler='torch' steps the cached engine; cached=False is available with the default sampler")
            return self._sample_torch_multinomial(eng, xs, c, cbs, start_loc, temperature, top_k_list, top_p_list)
        seed, offset = self._draw_rng(device, H * W * D)
        if not cached:
            return self._sample_uncached(eng, xs, c, cbs, start_loc, temperature, top_k_list, top_p_list, seed, offset)
        out = self._on_side_stream(device, lambda: eng.sample(xs, c, cbs, start_loc, temperature, top_k_list, top_p_list,
                                                              seed, offset, self.use_graph))
        return out

    def _sample_uncached(self, eng, xs, cond, cbs, start_loc, temperature, top_k_list, top_p_list, seed, offset):
        """``cached=False`` (transformers.py:352-356): nothing is carried from one step to the next -- every step recomputes the
        logits of the whole code map from the codes drawn so far (one teacher-forced pass of the engine per step, 256 per
        batch, as slow as the reference's own uncached loop) and samples position (h, w, d) from them with the draw the
        cached path would make at that step (same Philox key: seed, offset + step).  It exists, as in the reference, as the
        cross-check of the cache: the codes equal ``cached=True`` bit for bit (tests/test_gpu_parity.py).  (Both paths step the
        engine with B rows per launch -- the teacher-forced pass is the cached stepping over given codes, not one (B*H*W*D)-row
        pass -- so both pick the same kernels for a given B)"""
        from ... import _native
        (H, W, D) = self.block_size
        start = max(int(start_loc[0]) * W + int(start_loc[1]), 0)
        xs = xs.clone()
        for pos in range(start, H * W):
            h, w = divmod(pos, W)
            for d in range(D):
                logits = self._on_side_stream(xs.device, lambda: eng.logits(xs, cond, cbs))[:, h, w, d].contiguous()
                if self.vocab_size[d] < logits.shape[-1]:
                    logits[:, self.vocab_size[d]:] = float('-inf')          # LogitMask, as the sampling path applies it
                idx, _ = _native.sample_logits(logits, temperature, top_k_list[d], top_p_list[d], seed=seed,
                                               offset=offset + pos * D + d)
                xs[:, h, w, d] = idx
        return xs

    def _sample_torch_multinomial(self, eng, xs, cond, cbs, start_loc, temperature, top_k_list, top_p_list):
        """``self.sampler = 'torch'`` (or RQAMD_SAMPLER=torch): the loop of transformers.py:346-364 driven from the host, one engine
        step per (h, w, d), the draw by ``torch.multinomial(probs, num_samples=1)`` on the filtered probabilities -- the call
        sample_from_logits makes (rqvae/utils/utils.py:112), so the device generator is consumed exactly as the reference consumes
        it (one multinomial over a (B, V) tensor per step).  ~10x slower than the default on-device sampler (no graphs, one host
        round per step); meant for seed-for-seed comparisons, not for throughput."""
        from ... import _native
        (H, W, D) = self.block_size
        start = max(int(start_loc[0]) * W + int(start_loc[1]), 0)
        eng.step_begin(xs, cond, cbs)
        for pos in range(H * W):
            if pos < start:
                eng.step_logits(pos, -1)               # given codes: body KV cache only
                continue
            for d in range(D):
                logits = eng.step_logits(pos, d)
                _, probs = _native.sample_logits(logits, temperature, top_k_list[d], top_p_list[d], want_probs=True, want_samples=False)
                try:
                    idx = torch.multinomial(probs, num_samples=1).squeeze(-1)
                except RuntimeError:
                    print(probs, logits, torch.sum(probs), torch.sum(probs < 0))
                    raise
                eng.step_set_code(pos, d, idx)
        return eng.step_end()

    @staticmethod
    def _draw_rng(device, n_steps):
        """(seed, offset) of the device's default generator, advanced past this call: set_seed(seed + rank)
        in the drivers (main_sampling_fid.py:166-169) therefore makes sampling reproducible per rank."""
        if device.type == 'cuda':
            gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        else:
            gen = torch.default_generator
        seed = gen.initial_seed()
        try:
            offset = gen.get_offset()
            gen.set_offset(offset + 4 * ((n_steps + 3) // 4))
        except (RuntimeError, AttributeError):
            offset = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) * 4     # CPU generator (emulator tests)
        return seed, offset

    def compute_loss(self, logits, targets, use_soft_target=False):
        """transformers.py:371-382 (hard targets; soft targets are training-side)."""
        if use_soft_target:
            raise NotImplementedError('soft-target cross entropy (RQ-Transformer training) is out of scope')
        return F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1))
