"""CPU-only: host-side mirror logic that needs no device -- config handling, constructor errors,
state_dict tables, sampler argument normalisation."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import configs as C


def test_state_dict_tables_match_reference_shapes():
    from rqvae.models.rqvae import RQVAE
    from rqvae.models.rqtransformer import RQTransformer
    for hps, dd in (C.VAE_TINY, C.VAE_FFHQ):
        m = RQVAE(**hps, ddconfig=dd, checkpointing=False)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == \
            {k: tuple(s) for k, s in oracle.rqvae_param_shapes(hps, dd).items()}
        # shared codebook: one tensor under 4 names (quantizations.py:199-205)
        assert m.quantizer.codebooks[0] is m.quantizer.codebooks[3]
        assert not m.quantizer.codebooks[0].weight.requires_grad
    for cfg in (C.RQT_TINY, C.RQT_CC3M_654M):
        with torch.device('meta'):
            t = RQTransformer(cfg)
        assert {k: tuple(v.shape) for k, v in t.state_dict().items()} == \
            {k: tuple(s) for k, s in oracle.rqt_param_shapes(cfg).items()}


def test_constructor_errors_follow_the_reference():
    from rqvae.models.rqvae.quantizations import RQBottleneck
    from rqvae.models.rqtransformer import RQTransformer
    with pytest.raises(ValueError):
        RQBottleneck([8, 8, 256], [8, 8], 16)                       # quantizations.py:175-176
    with pytest.raises(ValueError):
        RQBottleneck([8, 8, 256], [3, 8, 4], 16)                    # :177-178
    with pytest.raises(ValueError):
        RQBottleneck([8, 8, 256], [8, 8, 4], [16] * 4, shared_codebook=True)   # :189-191
    bad = dict(C.RQT_TINY, block_size=[4, 4])
    with pytest.raises(ValueError):
        RQTransformer(bad)                                          # transformers.py:41-42
    rq = RQBottleneck([16, 16, 64], [8, 8, 4], 32)
    x = torch.arange(2 * 16 * 16 * 64, dtype=torch.float32).reshape(2, 16, 16, 64)
    assert rq.to_code_shape(x).shape == (2, 8, 8, 256)
    assert torch.equal(rq.to_latent_shape(rq.to_code_shape(x)), x)  # quantizations.py:216-235 round trip


def test_config_loading(tmp_path):
    from rqvae.utils.config import Config, augment_arch_defaults, load_config
    p = tmp_path / 'cfg.yaml'
    p.write_text('''
arch:
  type: rq-transformer
  block_size: [8, 8, 4]
  embed_dim: 1536
  input_embed_dim: 256
  shared_tok_emb: true
  shared_cls_emb: true
  input_emb_vqvae: true
  head_emb_vqvae: true
  cumsum_depth_ctx: true
  vocab_size_cond: 1000
  block_size_cond: 1
  vocab_size: 16384
  body: {n_layer: 42, block: {n_head: 24}}
  head: {n_layer: 6, block: {n_head: 24}}
''')
    cfg = load_config(str(p))
    arch = augment_arch_defaults(cfg.arch)
    assert arch.body.block.embed_dim == 1536 and arch.head.block.resid_pdrop == 0.1 and arch.body.block.gelu == 'v1'
    assert arch.block_size == [8, 8, 4] and arch.embd_pdrop == 0.0
    c2 = arch.copy()
    c2.body.n_layer = 1
    assert arch.body.n_layer == 42
    v = augment_arch_defaults(Config({'type': 'rq-vae', 'hparams': {'a': 1}}))
    # config.py:31-43 of the reference: the stage-1 defaults are merged UNDER the given config
    assert v.ema is None and v.checkpointing is False
    assert dict(**v.hparams) == {'loss_type': 'l1', 'restart_unused_codes': False, 'use_padding_idx': False, 'masked_dropout': 0.0, 'a': 1}
    v2 = augment_arch_defaults(Config({'type': 'rq-vae', 'hparams': {'loss_type': 'mse'}, 'checkpointing': True}))
    assert v2.hparams.loss_type == 'mse' and v2.checkpointing is True
    with pytest.raises(NotImplementedError):
        augment_arch_defaults(Config({'type': 'other'}))


def test_sample_argument_normalisation(monkeypatch):
    """transformers.py:314-330: top_k / top_p scalars, singletons and lists -> per-depth lists."""
    from rqvae.models.rqtransformer import RQTransformer
    ar = RQTransformer(C.RQT_TINY)
    seen = {}

    class FakeEngine:
        def sample(self, xs, c, cbs, start_loc, temperature, top_k, top_p, seed, offset, use_graph):
            seen.update(top_k=top_k, top_p=top_p, start=start_loc, t=temperature, cond=c)
            return xs.clone()

    class Aux:
        class quantizer:
            @staticmethod
            def codebook_list():
                return [torch.zeros((500, 64))] * 4
    picked = []
    monkeypatch.setattr(ar, '_eng', lambda amp=False: (picked.append(bool(amp)), FakeEngine())[1])
    z = torch.zeros((2, 4, 4, 4), dtype=torch.long)
    ar.sample(z, Aux)
    ar.sample(z, Aux, amp=True)
    assert picked == [False, True]                   # amp=True selects the fp16 engine (round 6)
    assert seen['top_k'] == [500] * 4 and seen['top_p'] == [1.0] * 4 and seen['cond'] is None
    ar.sample(z, Aux, top_k=1000, top_p=0.95, temperature=0.7, start_loc=(1, 2))
    assert seen['top_k'] == [500] * 4 and seen['top_p'] == [0.95] * 4 and seen['t'] == 0.7 and seen['start'] == (1, 2)
    ar.sample(z, Aux, top_k=[7], top_p=[1.5])
    assert seen['top_k'] == [7] * 4 and seen['top_p'] == [1.0] * 4
    ar.sample(z, Aux, top_k=[1, 2, 3, 4], top_p=[0.1, 0.2, 0.3, 0.4], cond=torch.tensor([3, 4]))
    assert seen['top_k'] == [1, 2, 3, 4] and seen['top_p'] == [0.1, 0.2, 0.3, 0.4] and seen['cond'].shape == (2, 1)
    with pytest.raises(AssertionError):
        ar.sample(torch.zeros((2, 8, 8, 4), dtype=torch.long), Aux)   # transformers.py:310
    with pytest.raises(ValueError):
        ar.sample(z, None)

    class BadAux:                                   # a codebook too small / too narrow for this transformer is refused
        class quantizer:
            @staticmethod
            def codebook_list():
                return [torch.zeros((100, 64))] * 4
    with pytest.raises(ValueError):
        ar.sample(z, BadAux)


def test_bench_support_count_strict_and_relaxed():
    """bench.py's post-run check: a sampled code must lie in the filtered support (top-k, top-p) of its teacher-forced logits; a code one
    token outside the strict top-p edge (bf16 logits through another kernel variant move the edge) is counted but tolerated by the
    relaxed filter (top-k + 2 %, top-p + 0.01); a code far outside fails both."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('rq_bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    V = 200
    # descending probabilities: token i has logit -0.05 i  ->  the top-p = 0.5 support is a prefix [0, n0)
    logits = np.tile((-0.05 * np.arange(V, dtype=np.float32))[None, None, None, None, :], (3, 1, 1, 1, 1))
    pr = oracle.filtered_probs(logits[:, 0, 0, 0], 1.0, 100, 0.5)
    n0 = int((pr[0] > 0).sum())
    pr2 = oracle.filtered_probs(logits[:, 0, 0, 0], 1.0, 102, 0.51)
    n1 = int((pr2[0] > 0).sum())
    assert 0 < n0 < n1 < 100
    codes = np.array([n0 - 1, n0, 150], dtype=np.int64).reshape(3, 1, 1, 1)       # inside / first token past the strict edge / far outside
    assert bench.count_outside_support(logits, codes, 100, 0.5) == (2, 1)
    assert bench.count_outside_support(logits[:2], codes[:2], 100, 0.5) == (1, 0)
    assert bench.count_outside_support(logits[:1], codes[:1], 100, 0.5) == (0, 0)
    # top-k edge: rank k is outside the strict filter, inside the relaxed one (k + 2 %)
    codes_k = np.array([99, 100, 101, 102], dtype=np.int64).reshape(4, 1, 1, 1)
    lg4 = np.tile(logits[:1], (4, 1, 1, 1, 1))
    assert bench.count_outside_support(lg4, codes_k, 100, None) == (3, 1)


def test_bench_gather_plan_and_multi_gpu_defaults():
    """bench.py: the pixel all-gather runs as ONE call while world x batch fits the gather budget and as
    slices above it -- and config.parallelism says which; the two models BASELINE.json quotes at a global batch on 8 GPUs default
    to its per-GPU share (64) when launched with more than one rank."""
    import bench
    per_img = 3 * 256 * 256 * 4
    assert bench.gather_plan(8, 64, per_img) == (64, 1)
    step, calls = bench.gather_plan(8, 10752, per_img)
    assert calls > 1 and step * 8 * per_img <= bench.GATHER_BUDGET_BYTES < (step + 1) * 8 * per_img and step * calls >= 10752
    assert bench.parallelism_note(1, 10752) == 'single GPU'
    assert 'one pixel all-gather' in bench.parallelism_note(8, 64)
    note = bench.parallelism_note(8, 10752)
    assert f'{calls} all-gathers' in note and 'keeps only the last gathered slice' in note
    assert bench.DEFAULT_BATCH_MULTI == {'xhuge': 64, 'txt3900m': 64} and 'huge' not in bench.DEFAULT_BATCH_MULTI


def test_read_ahead_torch_attributes():
    """RQVAE's read-ahead behind per-row calls (rqvae/models/rqvae/rqvae.py: _ReadAhead) recognises "row i of the batch the previous call
    saw" through two PRIVATE torch attributes: `Tensor._base` (a view's base tensor) and `Tensor._version` (bumped by in-place edits).
    If a torch release drops or changes either, the window logic must be revisited -- this test fails first."""
    import torch
    base = torch.zeros((4, 3))
    row = base[1:2]
    assert hasattr(row, '_base') and row._base is base and base._base is None
    v0 = base._version
    base[2].add_(1.0)
    assert base._version > v0 and row._version == base._version          # views share the version counter of their base
    from rqvae.models.rqvae import rqvae as rqvae_mod
    assert hasattr(rqvae_mod, '_ReadAhead')
    with torch.inference_mode():
        t = torch.zeros((2, 2))
    try:                                                                  # inference tensors have no version counter: the read-ahead steps aside
        t._version
        has_version = True
    except RuntimeError:
        has_version = False
    assert has_version is False
