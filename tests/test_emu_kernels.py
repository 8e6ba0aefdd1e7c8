"""CPU-only checks of the HIP kernel SOURCES through the host emulator (tests/emu): same .hip files,
same C ABI, same ctypes binding, executed by fibers instead of a GPU, compared against the oracle and
the reference-generated golden fixtures.  These tests exist because the build container has no GPU;
the authoritative parity tests are the `-m gpu` ones (tests/test_gpu_*.py)."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle import configs as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get('RQ_EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason='no host clang++ for the emulator build')


@pytest.fixture(scope='module')
def nat():
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
    import build_emu
    path = build_emu.build()
    from rqvae import _native
    import emu_binding
    saved = emu_binding.install(_native, path)
    yield _native
    emu_binding.restore(_native, saved)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_emu_rq_quantize_and_embed(nat, golden):
    g = golden('rq_small.npz')
    x, cb = T(g['x'].reshape(-1, 64)), T(g['codebook'])
    codes, quants = nat.rq_quantize(x, [cb] * 4)
    assert np.array_equal(codes.numpy().reshape(g['codes'].shape), g['codes'])
    np.testing.assert_array_equal(quants.numpy().reshape(g['quant_list'].shape), g['quant_list'])
    codes_only, none = nat.rq_quantize(x, [cb] * 4, want_quants=False)
    assert none is None and torch.equal(codes_only, codes)
    e0 = nat.rq_embed(codes, [cb] * 4, 0).numpy()
    np.testing.assert_array_equal(e0.reshape(g['quant_list'][-1].shape), g['quant_list'][-1])
    e1 = nat.rq_embed(codes, [cb] * 4, 1).numpy()
    np.testing.assert_array_equal(e1.reshape(g['embed_with_depth'].shape), g['embed_with_depth'])
    e2 = nat.rq_embed(codes, [cb] * 4, 2).numpy()
    np.testing.assert_array_equal(e2, np.cumsum(e1, 1, dtype=np.float32))


def test_emu_rq_quantize_ragged(nat):
    """n_vec not a multiple of the 64-vector tile, K not a multiple of the 128-code tile, unshared codebooks."""
    rng = np.random.default_rng(5)
    cbs = [rng.standard_normal((k, 128), dtype=np.float32) for k in (130, 70, 257)]
    x = rng.standard_normal((1, 5, 15, 128), dtype=np.float32)
    codes, quants = nat.rq_quantize(T(x.reshape(-1, 128)), [T(c) for c in cbs])
    oq, oc = oracle.rq_quantize(x, cbs)
    gaps, _ = oracle.rq_quantize_margins(x, cbs)
    assert gaps.min() > 1e-3
    assert np.array_equal(codes.numpy().reshape(oc.shape), oc)
    np.testing.assert_allclose(quants.numpy().reshape(3, 1, 5, 15, 128), np.stack(oq), rtol=0, atol=1e-6)
    assert nat.rq_quantize(T(np.zeros((0, 128), np.float32)), [T(c) for c in cbs])[0].shape == (0, 3)


def test_emu_rq_quantize_codebook_split(nat):
    """Few vectors, K >= 1024: the codebook is divided over blockIdx.y, one launch pair per depth (partial minima + combine).
    Must be bit-identical to the single-launch path (same vectors inside a batch large enough to take it) and to the oracle."""
    rng = np.random.default_rng(6)
    cbs = [rng.standard_normal((k, 64), dtype=np.float32) for k in (1100, 1100, 1300)]
    cbs[1] = cbs[0]                                                  # depths 0 / 1 share a codebook
    x = rng.standard_normal((70, 64), dtype=np.float32)
    codes, quants = nat.rq_quantize(T(x), [T(c) for c in cbs])                       # 2 tiles: split path
    oq, oc = oracle.rq_quantize(x.reshape(1, 1, 70, 64), cbs)
    gaps, _ = oracle.rq_quantize_margins(x.reshape(1, 1, 70, 64), cbs)
    assert gaps.min() > 1e-3
    assert np.array_equal(codes.numpy().reshape(oc.shape), oc)
    np.testing.assert_array_equal(quants.numpy().reshape(3, 1, 1, 70, 64), np.stack(oq))
    nat.dbg_set_row_scale(100)                 # the path choice sees 200 tiles: single launch over the same 70 vectors
    try:
        codes_b, quants_b = nat.rq_quantize(T(x), [T(c) for c in cbs])
    finally:
        nat.dbg_set_row_scale(1)
    assert torch.equal(codes_b, codes) and torch.equal(quants_b, quants)
    c2, none = nat.rq_quantize(T(x), [T(c) for c in cbs], want_quants=False)
    assert none is None and torch.equal(c2, codes)


def test_emu_rq_quantize_non_finite_rows(nat):
    """all-NaN / all-inf distance rows -> code 0 (torch.argmin's answer), never the 0x7fffffff seed; both launch forms."""
    rng = np.random.default_rng(15)
    for n_vec, K in ((200, 300), (70, 1100)):
        cb = rng.standard_normal((K, 64), dtype=np.float32)
        x = rng.standard_normal((n_vec, 64), dtype=np.float32)
        good, _ = nat.rq_quantize(T(x), [T(cb)] * 3)
        x[3, 7] = np.nan
        x[11, :] = np.inf
        codes, _ = nat.rq_quantize(T(x), [T(cb)] * 3)
        codes = codes.numpy()
        assert (codes[[3, 11]] == 0).all()
        keep = np.ones(n_vec, bool)
        keep[[3, 11]] = False
        assert np.array_equal(codes[keep], good.numpy()[keep])


@pytest.mark.parametrize('which', ['ragged', 'split'])
def test_emu_rq_quantize_codebook_dma_lands_late(nat, golden, monkeypatch, which):
    """Round 6: the quantiser's codebook ring (four 32-KB stages + the tile's norms, filled by LDS-DMA) with RQ_EMU_DMA=late -- every
    DMA lands only when the issuing lane's counted `s_waitcnt vmcnt(N)` retires it, so a fragment (or norm) read that is not behind the
    covering wait and the barrier returns stale bytes.  The default mode lands a DMA at issue (the worst case for a stage refilled while
    some wavefront still reads it); the three tests above run in that mode.  Depths 1, 2 and 4 chunks per tile, ragged K, the split form."""
    monkeypatch.setenv('RQ_EMU_DMA', 'late')
    if which == 'golden':
        test_emu_rq_quantize_and_embed(nat, golden)
    elif which == 'ragged':
        test_emu_rq_quantize_ragged(nat)
    else:
        test_emu_rq_quantize_codebook_split(nat)


def test_emu_rq_quantize_dims_and_tiny_codebooks(nat):
    """dim 192 / 256 (three / four chunks per tile: the released RQ-VAEs are 256) and codebooks smaller than one DMA chunk of norms
    (K = 2, 3, 5; K = 1: every code is 0), vs the oracle."""
    rng = np.random.default_rng(77)
    for dim, ks, n in ((192, (200, 129), 70), (256, (300, 128, 64), 130), (64, (2, 3, 5), 9)):
        cbs = [rng.standard_normal((k, dim), dtype=np.float32) for k in ks]
        x = rng.standard_normal((1, 1, n, dim), dtype=np.float32)
        codes, quants = nat.rq_quantize(T(x.reshape(-1, dim)), [T(c) for c in cbs])
        oq, oc = oracle.rq_quantize(x, cbs)
        gaps, _ = oracle.rq_quantize_margins(x, cbs)
        ok = gaps.reshape(n, len(ks)).min(1) > 1e-3
        assert ok.mean() > 0.9
        assert np.array_equal(codes.numpy().reshape(n, -1)[ok], oc.reshape(n, -1)[ok])
    one = rng.standard_normal((1, 64), dtype=np.float32)
    codes, quants = nat.rq_quantize(T(rng.standard_normal((5, 64), dtype=np.float32)), [T(one)] * 2)
    assert (codes.numpy() == 0).all() and np.array_equal(quants.numpy()[1], np.tile(one + one, (5, 1)))


def test_emu_rq_ema_update(nat, golden, fake_torch_rng):
    """Train-mode quantiser: the product's EMA codebook update + dead-code restart (csrc/quantize.hip: rq_ema_* kernels behind
    VQEmbedding.forward / RQBottleneck.quantize in train mode, quantizations.py:80-142,237-271) against the REFERENCE's outputs
    (tests/golden/rq_ema.npz).  Codes and gathered embeddings bit-exact; EMA statistics / refreshed weights to fp32 summation
    order (the reference sums a batch's vectors per code inside an sgemm, here in ascending vector order)."""
    from rqvae.models.rqvae.quantizations import RQBottleneck, VQEmbedding
    g = golden('rq_ema.npz')
    K, Dm, seed, decay = int(g['K']), int(g['D']), int(g['seed']), float(g['decay'])
    rng = np.random.default_rng(seed)
    cb = rng.standard_normal((K, Dm), dtype=np.float32)
    cs0 = rng.uniform(0.0, 3.0, K).astype(np.float32)
    xs = [rng.standard_normal((6, 8, 8, Dm), dtype=np.float32), rng.standard_normal((2, 8, 8, Dm), dtype=np.float32)]
    x_many = rng.standard_normal((384, Dm), dtype=np.float32)
    x_few = rng.standard_normal((128, Dm), dtype=np.float32)

    def load(vq):
        with torch.no_grad():
            vq.weight.copy_(T(np.concatenate([cb, np.zeros((1, Dm), np.float32)])))
            vq.embed_ema.copy_(T(cb * cs0[:, None]))
            vq.cluster_size_ema.copy_(T(cs0))
    rq = RQBottleneck([8, 8, Dm], [8, 8, 4], K, decay=decay, shared_codebook=True, restart_unused_codes=False).train()
    vq = rq.codebooks[0]
    load(vq)
    for b, x in enumerate(xs):
        quant_list, codes = rq.quantize(T(x))
        assert np.array_equal(_np(codes), g[f'codes{b}'])
        np.testing.assert_allclose(_np(quant_list[-1]), g[f'quant_last{b}'], rtol=0, atol=2e-6)
        for got, key in ((vq.weight[:-1], 'weight'), (vq.cluster_size_ema, 'cs'), (vq.embed_ema, 'ee')):
            np.testing.assert_allclose(_np(got), g[f'{key}{b}'], rtol=2e-5, atol=2e-6)
    # eval mode afterwards searches the refreshed codebook (cached ||c||^2 invalidated by the update)
    rq.eval()
    ql_e, codes_e = rq.quantize(T(xs[0]))
    oq, oc = oracle.rq_quantize(xs[0], [_np(vq.weight[:-1])] * 4)
    gaps, _ = oracle.rq_quantize_margins(xs[0], [_np(vq.weight[:-1])] * 4)
    clear = np.minimum.accumulate(gaps > 1e-3, axis=-1)
    assert np.array_equal(_np(codes_e)[clear], oc[clear]) and clear.mean() > 0.99
    for tag, xv, sd in (('many', x_many, seed + 1), ('few', x_few, seed + 2)):
        one = VQEmbedding(K, Dm, decay=decay, restart_unused_codes=True).train()
        load(one)
        fake_torch_rng(sd)
        emb, code = one(T(xv))
        assert np.array_equal(_np(code), g[f'{tag}_codes']) and np.array_equal(_np(emb), g[f'{tag}_embeds'])
        assert np.array_equal(_np(one.cluster_size_ema) == 1, g[f'{tag}_cs'] == 1)                 # the same codes restarted
        for got, key in ((one.weight[:-1], 'weight'), (one.cluster_size_ema, 'cs'), (one.embed_ema, 'ee')):
            np.testing.assert_allclose(_np(got), g[f'{tag}_{key}'], rtol=2e-5, atol=2e-6)
    # the kernel alone, ragged sizes: counts and sums vs numpy
    n_vec, Kk, Dd = 777, 45, 192
    xv = rng.standard_normal((n_vec, Dd), dtype=np.float32)
    idx = rng.integers(0, Kk, n_vec)
    idx[idx == 7] = 8                                                                              # an unused code
    count, vsum = nat.rq_ema_accumulate(T(xv), T(idx.astype(np.int64)), Kk)
    assert np.array_equal(_np(count), np.bincount(idx, minlength=Kk).astype(np.float32)) and _np(count)[7] == 0
    want = np.zeros((Kk, Dd), np.float64)
    np.add.at(want, idx, xv.astype(np.float64))
    np.testing.assert_allclose(_np(vsum), want, rtol=0, atol=2e-5)


def _np(t):
    return t.detach().cpu().numpy()


def test_emu_rq_soft_codes(nat, golden):
    """RQBottleneck.get_soft_codes against the reference fixture: softmax(-d / temp) per depth and codes."""
    g, gs = golden('rq_small.npz'), golden('rq_soft.npz')
    x, cb = T(g['x'][:1].reshape(-1, 64)), T(g['codebook'])
    norms = nat.rq_code_norms(cb)
    soft, codes = nat.rq_soft_codes(x, [cb] * 4, [norms] * 4, temp=float(gs['temp']))
    assert np.array_equal(codes.numpy().reshape(gs['codes'].shape), gs['codes'])
    ref = gs['soft'].reshape(-1, 4, 500)
    assert np.abs(soft.numpy() - ref).max() < 2e-5 and np.abs(soft.numpy().sum(-1) - 1).max() < 1e-5
    # stochastic codes: a draw from each depth's soft code -- must be a code with non-negligible probability; seeded
    s1, c1 = nat.rq_soft_codes(x, [cb] * 4, [norms] * 4, temp=200.0, stochastic=True, seed=5, offset=8)
    s2, c2 = nat.rq_soft_codes(x, [cb] * 4, [norms] * 4, temp=200.0, stochastic=True, seed=5, offset=8)
    assert torch.equal(c1, c2) and torch.equal(s1, s2)
    p = torch.gather(s1, 2, c1.unsqueeze(-1)).squeeze(-1)
    assert float(p.min()) > 1e-5 and not torch.equal(c1, codes)          # flat distribution at temp 200: draws differ from the argmin
    assert torch.equal(c1[:, 0] == codes[:, 0], c1[:, 0] == codes[:, 0])


@pytest.mark.parametrize('case', [3, 5, 6])
def test_emu_sampler_filter(nat, golden, case):
    g = golden('sampler.npz')
    t, k, p = g['cases'][case]
    _, probs = nat.sample_logits(T(g['logits']), t, None if k < 0 else int(k), None if p < 0 else float(p),
                                 want_probs=True, want_samples=False)
    ref = g[f'probs_{case}']
    o = probs.numpy()
    tie_free = [0, 2, 4, 5, 6, 7]
    assert 0.5 * np.abs(o - ref).sum(-1)[tie_free].max() < 1e-5
    assert np.abs(np.sort(o, -1) - np.sort(ref, -1)).sum(-1).max() < 1e-4
    assert np.array_equal((o > 0).sum(-1), (ref > 0).sum(-1))


def test_emu_sampler_heavy_ties(nat):
    """Rows made of 2-3 distinct logits: thousands of keys tie at the top-k threshold (more survivors than the
    compact tail holds, more prefix-sharing keys than one wavefront settles) and at the top-p boundary (the
    lowest-index rule).  Against the oracle's stable sort."""
    rng = np.random.default_rng(21)
    V = 4096
    logits = np.stack([rng.choice([0.5, 1.5], V, p=[0.4, 0.6]),              # 2 values, ~2450 at the top
                       rng.choice([-1.0, 0.0, 2.0], V, p=[0.5, 0.3, 0.2]),
                       np.full(V, 0.25)]).astype(np.float32)                  # constant row
    for k, p in ((10, None), (10, 0.9), (3000, 0.5), (None, 0.7)):
        _, probs = nat.sample_logits(T(logits), 1.0, k, p, want_probs=True, want_samples=False)
        ref = oracle.filtered_probs(logits, 1.0, k, p)
        o = probs.numpy()
        # thousands of EQUAL probabilities at the top-p boundary: how many of them fit under p depends on the fp32
        # summation order of the cumulative sum (torch, numpy and this kernel all differ) -> the kept count may be
        # off by one on such rows; everything else must agree exactly
        dn = np.abs((o > 0).sum(-1) - (ref > 0).sum(-1))
        assert dn.max() <= 1, (k, p, dn)
        for r in range(o.shape[0]):
            if dn[r] == 0:
                assert np.array_equal(o[r] > 0, ref[r] > 0) and np.abs(o[r] - ref[r]).max() < 1e-6, (k, p, r)
            else:
                assert ((o[r] > 0) != (ref[r] > 0)).sum() == 1 and 0.5 * np.abs(o[r] - ref[r]).sum() < 5e-4, (k, p, r)


def test_emu_sampler_draws(nat):
    """Draw statistics of the exponential-race sampler against the filtered distribution."""
    rng = np.random.default_rng(3)
    logits = np.tile((2.0 * rng.standard_normal((1, 40))).astype(np.float32), (64, 1))
    probs = oracle.filtered_probs(logits[:1], 1.0, 10, 0.9)[0]
    counts = np.zeros(40)
    for rep in range(3):
        s, _ = nat.sample_logits(T(logits), 1.0, 10, 0.9, seed=7, offset=rep)
        counts += np.bincount(s.numpy(), minlength=40)
    assert counts[probs == 0].sum() == 0
    n = counts.sum()
    chi2 = (((counts - n * probs) ** 2) / (n * probs + 1e-12))[probs > 0].sum()
    assert chi2 < 45.0        # 9 dof would be ~9; generous bound, catches a broken RNG/argmax
    s1, _ = nat.sample_logits(T(logits), 1.0, 10, 0.9, seed=7, offset=0)
    s2, _ = nat.sample_logits(T(logits), 1.0, 10, 0.9, seed=7, offset=0)
    assert torch.equal(s1, s2)
    # unfiltered draw (the reference's default top_k=None, top_p=None -> 1.0): single-pass Gumbel-max kernel
    probs = oracle.filtered_probs(logits[:1], 0.8, None, None)[0]
    counts = np.zeros(40)
    for rep in range(6):
        s, _ = nat.sample_logits(T(logits), 0.8, None, 1.0, seed=11, offset=rep)
        counts += np.bincount(s.numpy(), minlength=40)
    n = counts.sum()
    big = probs * n >= 5
    chi2 = (((counts - n * probs) ** 2) / (n * probs))[big].sum()
    assert chi2 < 3.0 * big.sum() + 20.0, chi2


def _rqt_engine(nat, cfg, params):
    eng = nat.RqtEngine(embed_dim=cfg['embed_dim'], n_head=cfg['body']['block']['n_head'], n_layer_body=cfg['body']['n_layer'],
                        n_layer_head=cfg['head']['n_layer'], vocab_size=cfg['vocab_size'], input_embed_dim=cfg['input_embed_dim'],
                        vocab_size_cond=cfg['vocab_size_cond'], block_size_cond=cfg['block_size_cond'],
                        block_size=cfg['block_size'], gelu_v2=cfg.get('gelu', 'v1') == 'v2', device='cpu')
    for k, v in params.items():
        eng.set_param(k, T(v))
    return eng


def test_emu_rqt_tiny_logits(nat, golden):
    g = golden('rqt_tiny.npz')
    cfg = C.RQT_TINY
    hps, dd = C.VAE_TINY
    cb = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['vae_seed']))['quantizer.codebooks.0.weight'][:-1]
    params = oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed']))
    eng = _rqt_engine(nat, cfg, params)
    codes, cond = T(g['codes'].astype(np.int64)), T(g['cond'].astype(np.int64))
    logits = eng.logits(codes, cond, [T(cb)] * 4).numpy()
    err = np.abs(logits - g['logits'])
    print('emu rqt tiny logits: max err %.4f mean err %.5f' % (err.max(), err.mean()))
    assert err.max() < 0.06 and err.mean() < 0.01          # bf16 weights/activations vs fp32 reference, |logits| <= 2.4
    # the large-batch kernel variants (LDS-DMA and register-blocked GEMM tiles, two heads per attention wavefront) are
    # selected by the row count; with the diagnostics factor they run on these 3 rows and must reproduce the result
    nat.dbg_set_row_scale(4096)
    try:
        logits_big = eng.logits(codes, cond, [T(cb)] * 4).numpy()
    finally:
        nat.dbg_set_row_scale(1)
    err2 = np.abs(logits_big - g['logits'])
    print('emu rqt tiny logits, large-batch variants: max err %.4f, max diff to the small-batch kernels %.5f'
          % (err2.max(), np.abs(logits_big - logits).max()))
    assert err2.max() < 0.06 and err2.mean() < 0.01
    assert np.abs(logits_big - logits).max() < 0.02


def test_emu_rqt_text_conditioned_logits(nat, golden):
    g = golden('rqt_tiny_txt.npz')
    cfg = C.RQT_TINY_TXT
    hps, dd = C.VAE_TINY
    cb = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['vae_seed']))['quantizer.codebooks.0.weight'][:-1]
    eng = _rqt_engine(nat, cfg, oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed'])))
    logits = eng.logits(T(g['codes'].astype(np.int64)), T(g['cond'].astype(np.int64)), [T(cb)] * 4).numpy()
    err = np.abs(logits - g['logits'])
    assert err.max() < 0.06 and err.mean() < 0.01
    # forward(): (seq_logits, cond_logits) -- cond_classifier over the prefix rows of the multi-token prefill
    params = oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed']))
    seq, cl = eng.forward(T(g['codes'].astype(np.int64)), T(g['cond'].astype(np.int64)), [T(cb)] * 4)
    assert np.array_equal(seq.numpy(), logits)
    ref = oracle.RQTransformerOracle(cfg, params).forward(g['codes'].astype(np.int64), [cb] * 4, g['cond'].astype(np.int64), return_cond_logits=True)
    assert np.abs(cl.numpy() - ref[1]).max() < 0.06


def test_emu_rqt_long_prefix(nat):
    """70 conditioning tokens: the prefill attention runs two query blocks (69 prefix tokens), the body context is
    16 + 69 = 85 keys (the 16-block DYN decode attention kernel); vs the oracle (pinned by the reference at 32 / 64 tokens)."""
    cfg = C.rqt(128, 2, 1, 1, 500, vocab_cond=20, block_cond=70, block_size=(4, 4, 4), input_embed_dim=64)
    params = oracle.make_params(oracle.rqt_param_shapes(cfg), 43)
    rng = np.random.default_rng(44)
    cb = rng.standard_normal((500, 64), dtype=np.float32)
    codes, cond = rng.integers(0, 500, (2, 4, 4, 4)), rng.integers(0, 20, (2, 70))
    eng = _rqt_engine(nat, cfg, params)
    seq, cl = eng.forward(T(codes), T(cond), [T(cb)] * 4)
    ref = oracle.RQTransformerOracle(cfg, params).forward(codes, [cb] * 4, cond, return_cond_logits=True)
    e1, e2 = np.abs(seq.numpy() - ref[0]).max(), np.abs(cl.numpy() - ref[1]).max()
    print('emu rqt long prefix: seq logits err %.4f, cond logits err %.4f' % (e1, e2))
    assert e1 < 0.06 and e2 < 0.06


@pytest.mark.parametrize('fmt', ['int8k', 'int8kv'])
def test_emu_rqt_int8k_key_cache(nat, golden, monkeypatch, fmt):
    """(fmt = int8kv, round 6: the VALUES of the body stack cached the same way as well -- bytes + a scale per (token, head), this token's own
    value bf16.)  Opt-in 8-bit key cache of the body stack (RQAMD_KV=int8k, read when an engine is created): cached keys as
    64 bytes + one fp32 absmax / 127 scale per (token, head), this token's own key and all values bf16 as before.  Through every
    attention kernel that has the variant -- <= 8 keys (attn_small), register blocks, the DYN long-context form behind a 69-token
    prefix, the quantising prefill -- against the
    reference's logits with the bound of the bf16 cache, and close to the bf16-cache engine; sampling stays inside the support of
    its own teacher-forced logits; an unknown format name is refused."""
    g = golden('rqt_tiny.npz')
    cfg = C.RQT_TINY
    hps, dd = C.VAE_TINY
    cb = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['vae_seed']))['quantizer.codebooks.0.weight'][:-1]
    params = oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed']))
    codes, cond = T(g['codes'].astype(np.int64)), T(g['cond'].astype(np.int64))
    base = _rqt_engine(nat, cfg, params).logits(codes, cond, [T(cb)] * 4).numpy()
    monkeypatch.setenv('RQAMD_KV', fmt)
    eng = _rqt_engine(nat, cfg, params)
    logits = eng.logits(codes, cond, [T(cb)] * 4).numpy()
    err = np.abs(logits - g['logits'])
    print(f'emu rqt tiny logits, {fmt} cache:'
          ' max err %.4f mean %.5f vs the reference; max %.4f mean %.5f vs the bf16 cache'
          % (err.max(), err.mean(), np.abs(logits - base).max(), np.abs(logits - base).mean()))
    assert err.max() < 0.06 and err.mean() < 0.01
    assert 0 < np.abs(logits - base).max() < 0.02            # the cache format is really in use, and costs little
    # (two heads per wavefront -- the large-batch form -- runs on the GPU: tests/test_gpu_parity_big.py::test_rqt_in1400m_int8k_key_cache)
    # sampling on it
    partial = torch.zeros((2, 4, 4, 4), dtype=torch.int64)
    out = eng.sample(partial, cond[:2].contiguous(), [T(cb)] * 4, (0, 0), 1.0, [5] * 4, [0.9] * 4, seed=11, offset=0, use_graph=False)
    tf = eng.logits(out, cond[:2].contiguous(), [T(cb)] * 4).numpy()
    for h in range(4):
        for w in range(4):
            for d in range(4):
                pr = oracle.filtered_probs(tf[:, h, w, d], 1.0, 5, 0.9)
                assert (pr[np.arange(2), out[:, h, w, d].numpy()] > 0).all()
    # text conditioning: quantising prefill (3 prefix tokens) and the 69-token prefix with the DYN decode kernel
    gt = golden('rqt_tiny_txt.npz')
    engt = _rqt_engine(nat, C.RQT_TINY_TXT, oracle.make_params(oracle.rqt_param_shapes(C.RQT_TINY_TXT), int(gt['seed'])))
    lt = engt.logits(T(gt['codes'].astype(np.int64)), T(gt['cond'].astype(np.int64)), [T(cb)] * 4).numpy()
    assert np.abs(lt - gt['logits']).max() < 0.06 and np.abs(lt - gt['logits']).mean() < 0.01
    cfgl = C.rqt(128, 2, 1, 1, 500, vocab_cond=20, block_cond=70, block_size=(4, 4, 4), input_embed_dim=64)
    pl = oracle.make_params(oracle.rqt_param_shapes(cfgl), 43)
    rng = np.random.default_rng(44)
    cbl = rng.standard_normal((500, 64), dtype=np.float32)
    cl_codes, cl_cond = rng.integers(0, 500, (2, 4, 4, 4)), rng.integers(0, 20, (2, 70))
    seq, _ = _rqt_engine(nat, cfgl, pl).forward(T(cl_codes), T(cl_cond), [T(cbl)] * 4)
    ref = oracle.RQTransformerOracle(cfgl, pl).forward(cl_codes, [cbl] * 4, cl_cond, return_cond_logits=True)
    print('emu rqt long prefix, 8-bit key cache: seq logits err %.4f' % np.abs(seq.numpy() - ref[0]).max())
    assert np.abs(seq.numpy() - ref[0]).max() < 0.06
    monkeypatch.setenv('RQAMD_KV', 'fp8')
    with pytest.raises(Exception):
        _rqt_engine(nat, cfg, params)


@pytest.mark.parametrize('tag', ['tuple', 'nocumsum', 'mixed', 'nobias', 'gelumix', 'heads', 'txtheads'])
def test_emu_rqt_flag_variants(nat, golden, tag):
    """primitives.py variants (TupleEmbedding + BatchLinear + per-depth vocabularies; cumsum_depth_ctx off; learned head
    embedding) through the mirror classes and the engine, against the reference's forward() logits.  'heads' / 'txtheads' (round 6):
    head sizes 32 / 128 / 16 and different head counts in the two stacks -- the plain attention kernels (decode step and prefix)."""
    from rqvae.models.rqtransformer import RQTransformer
    from rqvae.models.rqvae import RQVAE
    g = golden(f'rqt_var_{tag}.npz')
    cfg = {'tuple': C.RQT_TINY_TUPLE, 'nocumsum': C.RQT_TINY_NOCUMSUM, 'mixed': C.RQT_TINY_MIXED, 'nobias': C.RQT_TINY_NOBIAS,
           'gelumix': C.RQT_TINY_GELUMIX, 'heads': C.RQT_TINY_HEADS, 'txtheads': C.RQT_TINY_TXT_HEADS}[tag]
    hps, dd = C.VAE_TINY
    vae = RQVAE(**hps, ddconfig=dd, checkpointing=False)
    vae.load_state_dict({k: T(v) for k, v in oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['vae_seed'])).items()})
    ar = RQTransformer(cfg).eval()
    ar.load_state_dict({k: T(v) for k, v in oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed']), cfg).items()}, strict=True)
    codes, cond = T(g['codes'].astype(np.int64)), T(g['cond'].astype(np.int64))
    logits = ar(codes, vae if tag != 'tuple' else None, cond=cond)
    logits = (logits[0] if isinstance(logits, tuple) else logits).numpy()      # ('txtheads': (seq_logits, cond_logits))
    err = np.abs(logits - g['logits'])
    print(f'emu rqt variant {tag}: max err {err.max():.4f} mean {err.mean():.5f}')
    assert err.max() < 0.06 and err.mean() < 0.01
    if tag == 'tuple':          # (the GPU test samples all three variants)
        # TupleEmbedding.forward on its own (primitives.py:63-75 of the reference) through the library's gather: row code + offset_d
        emb = ar.tok_emb(codes)
        assert emb.shape == codes.shape + (ar.tok_emb.embedding_dim,)
        assert torch.equal(emb, ar.tok_emb.weight.detach()[codes + ar.tok_emb.offsets.view(1, 1, 1, -1)])
        # BatchLinear / LogitMask on their own (round 6: plain torch forms for stand-alone use, any device, differentiable)
        hvec = torch.randn(3, 4, 128)
        lin = ar.classifier.linear
        want = torch.einsum('bij,ijk->bik', hvec, lin.weight.detach()) + lin.bias.detach()
        assert torch.allclose(lin(hvec), want, atol=1e-5)
        assert torch.allclose(lin(hvec[:, :2], indices=[3, 1]), want[:, [3, 1]] - want[:, [3, 1]] + torch.einsum('bij,ijk->bik', hvec[:, :2], lin.weight.detach()[[3, 1]]) + lin.bias.detach()[[3, 1]], atol=1e-5)
        masked = ar.classifier.logit_mask(torch.zeros(2, 4, max(ar.vocab_size)))
        assert all(torch.isinf(masked[:, d, v:]).all() and not torch.isinf(masked[:, d, :v]).any() for d, v in enumerate(ar.vocab_size))
        ar.use_graph = False
        out = ar.sample(torch.zeros_like(codes), None, cond=cond, top_k=50, top_p=0.9)
        vs = ar.vocab_size
        assert all(int(out[..., d].max()) < vs[d] and int(out[..., d].min()) >= 0 for d in range(4))  # LogitMask: never beyond a depth's vocabulary


def test_emu_rqt_depth1_no_head_stack(nat):
    """head.n_layer = 0 with depth-1 codes: the "VQ-GAN" transformer shapes of the throughput script
    (measure_throughput/__main__.py:166-210); the classifier reads the body output + pos_emb_d directly."""
    cfg = C.rqt(128, 2, 2, 0, 500, vocab_cond=10, block_size=(4, 4, 1), input_embed_dim=64)
    params = oracle.make_params(oracle.rqt_param_shapes(cfg), 45)
    rng = np.random.default_rng(46)
    cb = rng.standard_normal((500, 64), dtype=np.float32)
    codes, cond = rng.integers(0, 500, (3, 4, 4, 1)), rng.integers(0, 10, (3, 1))
    eng = _rqt_engine(nat, cfg, params)
    logits = eng.logits(T(codes), T(cond), [T(cb)]).numpy()
    ref = oracle.RQTransformerOracle(cfg, params).forward(codes, [cb], cond)
    assert np.abs(logits - ref).max() < 0.06
    out = eng.sample(torch.zeros((3, 4, 4, 1), dtype=torch.int64), T(cond), [T(cb)], (0, 0), 1.0, [50], [0.9], seed=3, offset=0, use_graph=False)
    assert out.shape == (3, 4, 4, 1) and int(out.min()) >= 0 and int(out.max()) < 500


def test_emu_rqt_tiny_sample(nat, golden):
    """sample(): teacher-forcing the sampled codes back through the logits path must reproduce, at every
    step, a distribution under which the sampled code has non-zero filtered probability."""
    g = golden('rqt_tiny.npz')
    cfg = C.RQT_TINY
    hps, dd = C.VAE_TINY
    cb = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['vae_seed']))['quantizer.codebooks.0.weight'][:-1]
    params = oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed']))
    eng = _rqt_engine(nat, cfg, params)
    cond = T(g['cond'].astype(np.int64))
    cond = cond[:2].contiguous()
    partial = torch.zeros((2, 4, 4, 4), dtype=torch.int64)
    out = eng.sample(partial, cond, [T(cb)] * 4, (0, 0), 1.0, [5] * 4, [0.9] * 4, seed=11, offset=0, use_graph=False)
    assert out.shape == partial.shape and int(out.min()) >= 0 and int(out.max()) < cfg['vocab_size']
    assert int(partial.abs().sum()) == 0                      # input untouched (transformers.py:332 clones)
    logits = eng.logits(out, cond, [T(cb)] * 4).numpy()
    for h in range(4):
        for w in range(4):
            for d in range(4):
                pr = oracle.filtered_probs(logits[:, h, w, d], 1.0, 5, 0.9)
                sel = pr[np.arange(2), out[:, h, w, d].numpy()]
                assert (sel > 0).all()
    # (seed determinism, hipGraph == eager and start_loc are covered on the GPU: tests/test_gpu_parity.py)


def test_emu_rqt_stepping_form(nat, golden):
    """rqamd_rqt_step_*: the engine stepped one (position, depth) at a time with the codes supplied by the caller gives the
    logits of the teacher-forced pass bit for bit (same kernels, same order); a host loop that draws with torch.multinomial from
    the filtered probabilities (what RQTransformer.sampler = 'torch' does, the reference's sample_from_logits call) is
    reproducible under torch.manual_seed; start_loc skips positions."""
    g = golden('rqt_tiny.npz')
    cfg = C.RQT_TINY
    hps, dd = C.VAE_TINY
    cb = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['vae_seed']))['quantizer.codebooks.0.weight'][:-1]
    params = oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed']))
    eng = _rqt_engine(nat, cfg, params)
    cbs = [T(cb)] * 4
    cond = T(g['cond'].astype(np.int64))[:2].contiguous()
    codes = T(g['codes'].astype(np.int64))[:2].contiguous()
    want = eng.logits(codes, cond, cbs)
    eng.step_begin(torch.zeros_like(codes), cond, cbs)
    for pos in range(16):
        for d in range(4):
            lg = eng.step_logits(pos, d)
            assert torch.equal(lg, want[:, pos // 4, pos % 4, d]), (pos, d)
            eng.step_set_code(pos, d, codes[:, pos // 4, pos % 4, d].contiguous())
    assert torch.equal(eng.step_end(), codes)
    with pytest.raises(Exception):
        eng.step_logits(0, 0)                                  # no sequence in progress

    def host_sample(seed, start=0, partial=None):
        torch.manual_seed(seed)
        part = torch.zeros_like(codes) if partial is None else partial
        eng.step_begin(part, cond, cbs)
        for pos in range(16):
            if pos < start:
                eng.step_logits(pos, -1)
                continue
            for d in range(4):
                _, pr = nat.sample_logits(eng.step_logits(pos, d), 1.0, 5, 0.9, want_probs=True, want_samples=False)
                eng.step_set_code(pos, d, torch.multinomial(pr, num_samples=1).squeeze(-1))
        return eng.step_end()
    # start_loc = (2, 1): positions 0..8 keep the given codes (body passes only), the rest is drawn; same seed, same draw
    s, s2 = host_sample(5, start=9, partial=codes), host_sample(5, start=9, partial=codes)
    assert torch.equal(s, s2) and int(s.min()) >= 0 and int(s.max()) < cfg['vocab_size']
    assert torch.equal(s.reshape(2, 16, 4)[:, :9], codes.reshape(2, 16, 4)[:, :9])
    # (full-length draws, non-zero filtered probability of every drawn code, generator consumption: tests/test_gpu_parity.py)


def _vae_engine(nat, hps, dd, params):
    eng = nat.VaeEngine(dd, hps['embed_dim'], device='cpu')
    for k, v in params.items():
        if not k.startswith('quantizer.'):
            eng.set_param(k, T(v))
    return eng


def test_emu_vae_tiny(nat, golden):
    g = golden('vae_tiny.npz')
    hps, dd = C.VAE_TINY
    params = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['seed']))
    eng = _vae_engine(nat, hps, dd, params)
    cb = params['quantizer.codebooks.0.weight'][:-1]
    z_q = oracle.rq_embed_code(g['codes'], [cb] * 4)
    dec = eng.decode(T(z_q)).numpy()
    err = np.abs(dec - g['decode_code'])
    print('emu vae tiny decode: max err %.4f mean %.5f (|ref| max %.3f)' % (err.max(), err.mean(), np.abs(g['decode_code']).max()))
    assert err.max() < 0.05 and err.mean() < 0.008
    z_e = eng.encode(T(g['x'])).numpy()
    err = np.abs(z_e - g['z_e'])
    print('emu vae tiny encode: max err %.4f mean %.5f (|ref| max %.3f)' % (err.max(), err.mean(), np.abs(g['z_e']).max()))
    assert err.max() < 0.05 and err.mean() < 0.008


def test_emu_vae_tiny_without_resample_convs(nat, golden):
    """ddconfig.resamp_with_conv = False (layers.py:20-57 of the reference; round 6): Upsample is the bare nearest x2, Downsample a 2 x 2
    average pool -- rqamd_vae_set_option(h, "resamp_with_conv", 0), no *.upsample.conv / *.downsample.conv parameters -- against the
    reference's own outputs for that config."""
    g = golden('vae_tiny_noresamp.npz')
    hps, dd = C.VAE_TINY_NORESAMP
    params = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['seed']))
    assert not any('sample.conv' in k for k in params)
    eng = _vae_engine(nat, hps, dd, params)
    cb = params['quantizer.codebooks.0.weight'][:-1]
    dec = eng.decode(T(oracle.rq_embed_code(g['codes'], [cb] * 4))).numpy()
    err = np.abs(dec - g['decode_code'])
    print('emu vae tiny (no resample convs) decode: max err %.4f mean %.5f' % (err.max(), err.mean()))
    assert err.max() < 0.05 and err.mean() < 0.008
    err = np.abs(eng.encode(T(g['x'])).numpy() - g['z_e'])
    print('emu vae tiny (no resample convs) encode: max err %.4f mean %.5f' % (err.max(), err.mean()))
    assert err.max() < 0.05 and err.mean() < 0.008


def test_emu_vae_batch_invariance_across_split_k_forms(nat, golden):
    """An image's pixels / latents do not depend on how many images shared its call: calls of <= 8 images divide the K loop of
    the low-resolution convs over workgroups (fp32 slabs + splitk_reduce), larger calls fold the same chunks inside one
    workgroup (GemmArgs::vsplit) -- the same additions in the same order, so rows of a 10-image call must equal the same
    images decoded / encoded one, three and eight at a time BIT FOR BIT (the speculative batching behind
    RQVAE.decode_code hands out rows of a batched decode in place of per-image calls)."""
    g = golden('vae_tiny.npz')
    hps, dd = C.VAE_TINY
    params = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['seed']))
    eng = _vae_engine(nat, hps, dd, params)
    rng = np.random.default_rng(12)
    cb = params['quantizer.codebooks.0.weight'][:-1]
    codes = rng.integers(0, hps['n_embed'], (10, 8, 8, 4))
    z_q = T(oracle.rq_embed_code(codes, [cb] * 4))
    big = eng.decode(z_q).numpy()                                   # 10 images: virtual split-K
    for lo, hi in ((0, 1), (3, 4), (1, 4), (2, 10), (9, 10)):       # 1 / 1 / 3 / 8 / 1 images: real split-K (graph replay up to 4)
        np.testing.assert_array_equal(eng.decode(z_q[lo:hi].contiguous()).numpy(), big[lo:hi])
    x = T(np.clip(rng.standard_normal((10, 3, 16, 16), dtype=np.float32), -1, 1))
    zbig = eng.encode(x).numpy()
    for lo, hi in ((0, 1), (4, 7), (2, 10)):
        np.testing.assert_array_equal(eng.encode(x[lo:hi].contiguous()).numpy(), zbig[lo:hi])


def test_emu_vae_decode_code_read_ahead(nat, golden):
    """RQVAE.decode_code / RQVAE.forward called one row at a time on views of a batch (the reference drivers' loops,
    measure_throughput/__main__.py:297-299, main_sampling_fid.py:223, rqvae/metrics/fid.py:167-169) are served from batched
    passes over the rows that follow; every row equals the cold single-image call bit for bit, and nothing stale is ever
    served.  (Sized for the emulator: 9 rows = one cold call + one window of 8; the GPU tests run 150.)"""
    from rqvae.models.rqvae import RQVAE
    g = golden('vae_tiny.npz')
    hps, dd = C.VAE_TINY
    params = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['seed']))
    vae = RQVAE(**hps, ddconfig=dd, checkpointing=False)
    vae.load_state_dict({k: T(v) for k, v in params.items()})
    vae.eval()
    rng = np.random.default_rng(21)
    codes = T(rng.integers(0, hps['n_embed'], (9, 8, 8, 4)))
    cold = {i: vae.decode_code(codes[i:i + 1].clone()) for i in (0, 1, 6, 8)}       # not views: one engine call each
    st = vae._ahead
    assert st.engine_calls == 0 and st.hits == 0
    pixels = torch.cat([vae.decode_code(chunk) for chunk in codes.chunk(9)], dim=0)    # measure_throughput/__main__.py:297-299
    assert st.engine_calls == 2 and st.hits == 7                                    # 1 row cold, then a window of 8
    for i, c in cold.items():
        assert torch.equal(pixels[i:i + 1], c), i
    # an edit of the codes (version counter) is seen: nothing stale
    calls = st.engine_calls
    codes[6] = codes[0]
    assert torch.equal(vae.decode_code(codes[6:7]), cold[0]) and st.engine_calls == calls + 1
    # an in-place edit of a served window is never handed out again
    w = vae.decode_code(codes[1:2])
    assert torch.equal(w, cold[1])
    w.add_(1.0)
    assert torch.equal(vae.decode_code(codes[1:2]), cold[1])
    # a different tensor object over equal contents starts cold (storage addresses can be recycled)
    other = codes.clone()
    calls = st.engine_calls
    assert torch.equal(vae.decode_code(other[8:9]), cold[8]) and st.engine_calls == calls + 1
    # a weight edit invalidates the window
    vae.decode_code(other[0:1])
    with torch.no_grad():
        vae.decoder.conv_out.bias.add_(0.5)
    np.testing.assert_allclose(vae.decode_code(other[1:2]).numpy(), cold[1].numpy() + 0.5, rtol=0, atol=1e-5)     # not the stale window
    with torch.no_grad():
        vae.decoder.conv_out.bias.sub_(0.5)
    # RQAMD_DECODE_AHEAD=0 / max_rows = 0 switches the read-ahead off
    hits = st.hits
    st.max_rows = 0
    vae.decode_code(codes[0:1])
    vae.decode_code(codes[1:2])
    assert st.hits == hits
    # the rFID loop (rqvae/metrics/fid.py:167-169): stage1_model(imgs[i:i+1])[0] on row views of an image batch
    x = T(np.clip(rng.standard_normal((9, 3, 16, 16), dtype=np.float32), -1, 1))
    cold_f = {i: vae(x[i:i + 1].clone()) for i in (0, 8)}
    sf = vae._ahead_fwd
    assert sf.engine_calls == 0
    rows_f = [vae(x[i:i + 1]) for i in range(9)]
    assert sf.engine_calls == 2 and sf.hits == 7
    for i, (o0, l0, c0) in cold_f.items():
        o, l, c = rows_f[i]
        assert torch.equal(o, o0) and torch.equal(c, c0) and torch.equal(l, l0) and l.shape == l0.shape


def test_emu_vae_read_ahead_inference_mode_and_in_place_edits(nat, golden):
    """(1) the per-row driver loop under torch.inference_mode() -- inference tensors have no version counter, so the
    read-ahead must step aside instead of raising; (2) a caller that edits every row it is handed in place
    (decode_code(c[i:i+1]).clamp_()) gets correct rows and, once the first edit of a served window has been seen, one engine
    call per row for the rest of that batch -- not a ramped-up window recomputed at every call."""
    from rqvae.models.rqvae import RQVAE
    g = golden('vae_tiny.npz')
    hps, dd = C.VAE_TINY
    params = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['seed']))
    vae = RQVAE(**hps, ddconfig=dd, checkpointing=False)
    vae.load_state_dict({k: T(v) for k, v in params.items()})
    vae.eval()
    rng = np.random.default_rng(22)
    codes = T(rng.integers(0, hps['n_embed'], (6, 8, 8, 4)))
    cold = [vae.decode_code(codes[i:i + 1].clone()) for i in range(6)]
    st = vae._ahead
    with torch.inference_mode():
        rows = [vae.decode_code(codes[i:i + 1]) for i in range(6)]                   # base created outside, results inside
        inner = codes.clone()                                                      # an inference tensor as the batch
        rows2 = [vae.decode_code(inner[i:i + 1]) for i in range(3)]
    assert st.hits == 0
    for i in range(6):
        assert torch.equal(rows[i], cold[i])
    for i in range(3):
        assert torch.equal(rows2[i], cold[i])
    calls = st.engine_calls
    out = []
    for i in range(6):
        r = vae.decode_code(codes[i:i + 1])
        assert torch.equal(r, cold[i]), i
        out.append(r.clamp_(0, 1))                                                   # in-place edit of the served row
    # row 0 cold, rows 1.. one window of 5 (read ahead), row 2 sees the edited window: from there on one call per row
    assert st.engine_calls - calls == 2 + 4, st.engine_calls - calls
    for i in range(6):
        assert torch.equal(out[i], cold[i].clamp(0, 1))


def test_emu_gemm_tiles_and_lds_dma(nat):
    """Decode-step GEMM through the diagnostics entry: register-staged and LDS-DMA staged (2 / 3 stages) operand paths,
    ragged M / N (clamped rows), odd and even K-tile counts, bf16 / fp32 / split-K epilogues, vs fp32 matmul."""
    rng = np.random.default_rng(17)
    for (M, N, K) in ((200, 192, 320), (128, 64, 64), (130, 70, 448)):
        a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((0.1 * rng.standard_normal((N, K))).astype(np.float32)).to(torch.bfloat16)
        bias = T(rng.standard_normal(N).astype(np.float32))
        ref = a.float().numpy() @ w.float().numpy().T + bias.numpy()
        for gl in (0, 64, 96):
            # tile codes 257x128 / 129x128 = the half-depth-stage kernel (K-steps of 32; LDS-DMA only): 4 waves of 128x64 / 64x64
            for (bm, bn) in ((128, 64), (128, 128), (256, 128)) + (((257, 128),) if gl else ()):
                out = nat.dbg_gemm(a, w, bias, epi=3 + gl, bm=bm, bn=bn, splitk=1).numpy()
                assert np.abs(out - ref).max() < 2e-3 * np.abs(ref).max(), (M, N, K, gl, bm, bn)
            for (bm, bn) in ((128, 64),) + (((257, 128),) if gl else ()):
                out = nat.dbg_gemm(a, w, bias, epi=0 + gl, bm=bm, bn=bn, splitk=1).float().numpy()
                assert np.abs(out - ref).max() < 1e-2 * np.abs(ref).max(), (M, N, K, gl, bm, bn)
                if K >= 128:
                    out = nat.dbg_gemm(a, w, None, epi=4 + gl, bm=bm, bn=bn, splitk=2).numpy().sum(0)
                    assert np.abs(out - (ref - bias.numpy())).max() < 2e-3 * np.abs(ref).max(), (M, N, K, gl, bm, bn)


def test_emu_gemm_256x256_eight_phase(nat):
    """256 x 256 eight-phase kernel (gemm_p8_kernel): unit layout / swizzle / fragment maps, the staggered barrier count
    (wave row 1 runs one barrier behind and the counts must match at the end), partial M / N tiles, the shortened waits of
    the last two K-tiles, split-K, and all epilogue families.  The emulator executes an LDS-DMA at issue, so it cannot see a
    RAW hazard (a read before the data landed); a WAR hazard (a unit refilled while some wave still has to read it) shows up
    as wrong results for the fiber interleavings the emulator produces."""
    rng = np.random.default_rng(17)
    for (M, N, K) in ((300, 512, 256), (256, 300, 128), (520, 256, 384)):
        a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((0.1 * rng.standard_normal((N, K))).astype(np.float32)).to(torch.bfloat16)
        bias = T(rng.standard_normal(N).astype(np.float32))
        ref = a.float().numpy() @ w.float().numpy().T + bias.numpy()
        out = nat.dbg_gemm(a, w, bias, epi=3, bm=256, bn=256, splitk=1).numpy()              # fp32 rows (TR = 0)
        assert np.abs(out - ref).max() < 2e-3 * np.abs(ref).max(), (M, N, K)
        out = nat.dbg_gemm(a, w, bias, epi=0, bm=256, bn=256, splitk=1).float().numpy()      # bf16 through the LDS transpose (TR = 1)
        assert np.abs(out - ref).max() < 1e-2 * np.abs(ref).max(), (M, N, K)
        # two phases per K-tile (+ 512) against four (+ 1024): the same MFMA order per accumulator, so the same bits
        for epi in (3, 0, 4):
            bias_ = None if epi == 4 else bias
            assert torch.equal(nat.dbg_gemm(a, w, bias_, epi=epi + 512, bm=256, bn=256, splitk=1),
                               nat.dbg_gemm(a, w, bias_, epi=epi + 1024, bm=256, bn=256, splitk=1)), (M, N, K, epi)
        # GELU family (compile-time EK = 1 / 6 in the 256 x 256 kernel; interior tiles take the check-free epilogue) against the
        # 128 x 64 kernel's run-time form of the same epilogue
        g256 = nat.dbg_gemm(a, w, bias, epi=1, bm=256, bn=256, splitk=1).float().numpy()
        g128 = nat.dbg_gemm(a, w, bias, epi=1, bm=128, bn=64, splitk=1).float().numpy()
        assert np.abs(g256 - g128).max() <= 2e-2 * np.abs(g128).max(), (M, N, K)
        gref = 0.5 * ref * (1.0 + np.vectorize(__import__('math').erf)(ref / np.sqrt(2.0)))
        assert np.abs(g256 - gref).max() < 1.5e-2 * np.abs(gref).max(), (M, N, K)
        # residual-stream epilogue (4 + 2048): out = (out + a w^T) + bias in place, the additions in the order of slab + resid_ln
        x0 = T(rng.standard_normal((M, N)).astype(np.float32))
        slab = nat.dbg_gemm(a, w, None, epi=4, bm=256, bn=256, splitk=1)[0]
        xs = x0.clone()
        if N % 4 == 0:
            nat.dbg_gemm(a, w, bias, epi=4 + 2048, bm=256, bn=256, splitk=1, out=xs)
            assert torch.equal(xs, (x0 + slab) + bias), (M, N, K)
            xs2 = x0.clone()
            nat.dbg_gemm(a, w, bias, epi=4 + 2048, bm=128, bn=64, splitk=1, out=xs2)               # the shared epilogue of the other tiles
            assert np.abs(xs2.numpy() - xs.numpy()).max() < 1e-4 * np.abs(xs.numpy()).max()
        if K >= 256:
            out = nat.dbg_gemm(a, w, None, epi=4, bm=256, bn=256, splitk=2).numpy().sum(0)   # split-K slabs, 2 K-tiles per split
            assert np.abs(out - (ref - bias.numpy())).max() < 2e-3 * np.abs(ref).max(), (M, N, K)
    with pytest.raises(NotImplementedError):
        nat.dbg_gemm(a, w, bias, epi=3, bm=256, bn=256, splitk=0 + 0, out=None) if False else nat.dbg_gemm(
            a[:, :64].contiguous(), w[:, :64].contiguous(), bias, epi=3, bm=256, bn=256, splitk=1)   # a single K-tile is refused


def test_emu_gemm_mid_batch_tiles(nat):
    """Round 5: the eight- / sixteen-wavefront LDS-DMA tiles of the 129 .. 2047 row decode step (tile codes 132 x {64, 192},
    136 x {128, 256}, 264 x 128; three ring stages): every epilogue family incl. K splits and the in-place residual update, ragged M / N,
    and the engine's own choice (bm = bn = 0) for row counts in that range -- vs fp32 matmul, and auto == the explicit launch."""
    rng = np.random.default_rng(29)
    for (M, N, K) in ((200, 200, 448), (300, 132, 640)):
        a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((0.1 * rng.standard_normal((N, K))).astype(np.float32)).to(torch.bfloat16)
        bias = T(rng.standard_normal(N).astype(np.float32))
        ref = a.float().numpy() @ w.float().numpy().T + bias.numpy()
        scale = np.abs(ref).max()
        for (bm, bn) in ((132, 64), (136, 128), (132, 192), (264, 128), (136, 256)):
            out = nat.dbg_gemm(a, w, bias, epi=3 + 96, bm=bm, bn=bn, splitk=1).numpy()
            assert np.abs(out - ref).max() < 2e-3 * scale, (M, N, K, bm, bn)
            out = nat.dbg_gemm(a, w, bias, epi=0 + 96, bm=bm, bn=bn, splitk=1).float().numpy()
            assert np.abs(out - ref).max() < 1e-2 * scale, (M, N, K, bm, bn)
            out = nat.dbg_gemm(a, w, None, epi=4 + 96, bm=bm, bn=bn, splitk=K // 64 // 2 if K == 448 else 2).numpy().sum(0)
            assert np.abs(out - (ref - bias.numpy())).max() < 2e-3 * scale, (M, N, K, bm, bn)
            x0 = T(rng.standard_normal((M, N)).astype(np.float32))
            xs = x0.clone()
            nat.dbg_gemm(a, w, bias, epi=4 + 96 + 2048, bm=bm, bn=bn, splitk=1, out=xs)
            slab = nat.dbg_gemm(a, w, None, epi=4 + 96, bm=bm, bn=bn, splitk=1)[0]
            assert torch.equal(xs, (x0 + slab) + bias), (M, N, K, bm, bn)
    # round 6: from four m-tiles up the slab GEMMs deal K SLICES to groups of XCDs (schedule 4 in gemm.h: one-dimensional grid, the slab index
    # comes from the workgroup id) -- every slab holds exactly its K range, for 2 / 4 / 8 slices, ragged M and N, more n-tiles than XCDs in a group
    M, N, K = 520, 328, 1024
    a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(torch.bfloat16)
    w = torch.from_numpy((0.1 * rng.standard_normal((N, K))).astype(np.float32)).to(torch.bfloat16)
    for (bm, bn) in ((132, 64), (136, 128)):
        for sk in (2, 4, 8):
            slabs = nat.dbg_gemm(a, w, None, epi=4 + 96, bm=bm, bn=bn, splitk=sk).numpy()
            kk = K // sk
            for z in range(sk):
                ref = a[:, z * kk:(z + 1) * kk].float().numpy() @ w[:, z * kk:(z + 1) * kk].float().numpy().T
                assert np.abs(slabs[z] - ref).max() < 2e-3 * np.abs(ref).max(), (bm, bn, sk, z)
    # the engine's own tile choice in the mid range: a slab GEMM (K split allowed), a bf16 GEMM and wide fp32 rows
    for (M, N, K, epi) in ((200, 256, 1024, 4), (500, 384, 256, 0), (300, 8192, 128, 3)):
        a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((0.1 * rng.standard_normal((N, K))).astype(np.float32)).to(torch.bfloat16)
        bias = T(rng.standard_normal(N).astype(np.float32))
        ref = a.float().numpy() @ w.float().numpy().T + (0 if epi == 4 else bias.numpy())
        out = nat.dbg_gemm(a, w, None if epi == 4 else bias, epi=epi, bm=0, bn=0, splitk=0)
        got = out.float().numpy()
        if epi == 4:
            # (the diagnostics entry sizes the slab buffer for 8 splits and the picker may use fewer: sum the written ones)
            full = torch.zeros((8, M, N))
            nat.dbg_gemm(a, w, None, epi=epi, bm=0, bn=0, splitk=0, out=full)
            got = full.numpy().sum(0)
        assert np.abs(got - ref).max() < (1e-2 if epi == 0 else 2e-3) * np.abs(ref).max(), (M, N, K, epi)


@pytest.mark.parametrize('which', ['tiles', 'eight_phase', 'stream', 'mid'])
def test_emu_gemm_lds_dma_lands_late(nat, monkeypatch, which):
    """The LDS-DMA GEMMs again with RQ_EMU_DMA=late: a DMA lands only when the issuing lane's counted `s_waitcnt vmcnt(N)` retires it --
    the latest moment the hardware allows -- so a fragment read that is not ordered behind the covering wait (+ a barrier for other
    wavefronts' data) returns stale LDS bytes and the result is wrong.  The default mode lands a DMA at issue (the worst case for WAR);
    a schedule has to pass in both."""
    monkeypatch.setenv('RQ_EMU_DMA', 'late')
    if which == 'tiles':
        test_emu_gemm_tiles_and_lds_dma(nat)
    elif which == 'eight_phase':
        test_emu_gemm_256x256_eight_phase(nat)
    elif which == 'mid':
        test_emu_gemm_mid_batch_tiles(nat)
    else:
        test_emu_gemm_stream(nat)


def test_emu_gemm_skinny(nat):
    """M <= 64 skinny kernel: k-permuted operand fragments, in-workgroup split-K over eight wavefronts, LDS reduction, global
    split-K slabs, ragged M / N, every epilogue family."""
    rng = np.random.default_rng(19)
    for (M, N, K) in ((64, 96, 512), (37, 70, 1024), (1, 64, 512)):
        a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((0.1 * rng.standard_normal((N, K))).astype(np.float32)).to(torch.bfloat16)
        bias = T(rng.standard_normal(N).astype(np.float32))
        ref = a.float().numpy() @ w.float().numpy().T + bias.numpy()
        out = nat.dbg_gemm(a, w, bias, epi=3, bm=64, bn=32, splitk=1).numpy()
        assert np.abs(out - ref).max() < 2e-3 * np.abs(ref).max(), (M, N, K)
        out = nat.dbg_gemm(a, w, bias, epi=0, bm=64, bn=32, splitk=1).float().numpy()
        assert np.abs(out - ref).max() < 1e-2 * np.abs(ref).max(), (M, N, K)
        if K >= 1024:
            out = nat.dbg_gemm(a, w, None, epi=4, bm=64, bn=32, splitk=2).numpy().sum(0)
            assert np.abs(out - (ref - bias.numpy())).max() < 2e-3 * np.abs(ref).max(), (M, N, K)


def test_emu_gemm_stream(nat):
    """Small-batch weight-streaming kernel (gemm_stream_kernel: four barrier-free wavefronts with private LDS-DMA rings, every
    fourth K-tile each, LDS reduction): 64- and 128-row forms, ragged M / N, several m-tiles, K-tile counts that leave
    wavefronts with unequal (or no) work, every epilogue family incl. split-K slabs and the in-place residual update; the two
    forms agree bit for bit on shared rows."""
    rng = np.random.default_rng(23)
    for (M, N, K) in ((64, 96, 512), (37, 70, 1536), (1, 64, 128), (128, 64, 384), (200, 100, 640)):
        a = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((0.1 * rng.standard_normal((N, K))).astype(np.float32)).to(torch.bfloat16)
        bias = T(rng.standard_normal(N).astype(np.float32))
        ref = a.float().numpy() @ w.float().numpy().T + bias.numpy()
        scale = np.abs(ref).max()
        outs = {}
        for bm in (66, 130):
            out = nat.dbg_gemm(a, w, bias, epi=3, bm=bm, bn=32, splitk=1).numpy()
            assert np.abs(out - ref).max() < 2e-3 * scale, (M, N, K, bm)
            outs[bm] = out
            out = nat.dbg_gemm(a, w, bias, epi=0, bm=bm, bn=32, splitk=1).float().numpy()
            assert np.abs(out - ref).max() < 1e-2 * scale, (M, N, K, bm)
            if K >= 512:
                out = nat.dbg_gemm(a, w, None, epi=4, bm=bm, bn=32, splitk=2).numpy().sum(0)
                assert np.abs(out - (ref - bias.numpy())).max() < 2e-3 * scale, (M, N, K, bm)
            if N % 4 == 0:
                x0 = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
                xs = x0.clone()
                nat.dbg_gemm(a, w, bias, epi=4 + 2048, bm=bm, bn=32, splitk=1, out=xs)
                slab = nat.dbg_gemm(a, w, None, epi=4, bm=bm, bn=32, splitk=1)[0]
                assert torch.equal(xs, (x0 + slab) + bias), (M, N, K, bm)
        assert np.array_equal(outs[66], outs[130]), (M, N, K)
        if M <= 64:
            # 64-row weight tiles (round 4: GEMMs too wide for one round of 32-row tiles): the same arithmetic per output element
            assert np.array_equal(nat.dbg_gemm(a, w, bias, epi=3, bm=66, bn=64, splitk=1).numpy(), outs[66]), (M, N, K)
            assert torch.equal(nat.dbg_gemm(a, w, bias, epi=0, bm=66, bn=64, splitk=1), nat.dbg_gemm(a, w, bias, epi=0, bm=66, bn=32, splitk=1))
            if K >= 512:
                assert torch.equal(nat.dbg_gemm(a, w, None, epi=4, bm=66, bn=64, splitk=2), nat.dbg_gemm(a, w, None, epi=4, bm=66, bn=32, splitk=2))
            if N % 4 == 0:
                x64 = x0.clone()
                nat.dbg_gemm(a, w, bias, epi=4 + 2048, bm=66, bn=64, splitk=1, out=x64)
                assert torch.equal(x64, xs), (M, N, K)
    # the engine's own tile choice (bm = bn = 0) for a 64-row GEMM wider than one round of 32-row tiles (N / 32 > 256: fc1 of the
    # E = 2560 models, the classifier): 64-row weight tiles, ragged last tile -- same bits as the 32-row tiles
    a = torch.from_numpy(rng.standard_normal((5, 128)).astype(np.float32)).to(torch.bfloat16)
    w = torch.from_numpy((0.1 * rng.standard_normal((8240, 128))).astype(np.float32)).to(torch.bfloat16)
    bias = T(rng.standard_normal(8240).astype(np.float32))
    auto = nat.dbg_gemm(a, w, bias, epi=3, bm=0, bn=0, splitk=0).numpy()
    assert np.abs(auto - (a.float().numpy() @ w.float().numpy().T + bias.numpy())).max() < 2e-3 * np.abs(auto).max()
    assert np.array_equal(auto, nat.dbg_gemm(a, w, bias, epi=3, bm=66, bn=32, splitk=1).numpy())
    # ... and the K-split branch of the same picker: a slab GEMM of the wide models (proj at E = 2560: N / 32 x 4 > 256
    # workgroups, N / 64 x 4 <= 256) takes 64-row weight tiles with FOUR K slices -- the explicit launch, bit for bit, and the slabs
    # sum to the fp32 product
    a = torch.from_numpy(rng.standard_normal((5, 2560)).astype(np.float32)).to(torch.bfloat16)
    w = torch.from_numpy((0.05 * rng.standard_normal((2560, 2560))).astype(np.float32)).to(torch.bfloat16)
    auto = torch.zeros((8, 5, 2560))
    nat.dbg_gemm(a, w, None, epi=4, bm=0, bn=0, splitk=0, out=auto)
    assert float(auto[4:].abs().max()) == 0.0 and float(auto[3].abs().max()) > 0.0          # exactly four slabs were written
    assert torch.equal(auto[:4], nat.dbg_gemm(a, w, None, epi=4, bm=66, bn=64, splitk=4))
    ref = a.float().numpy() @ w.float().numpy().T
    assert np.abs(auto.numpy().sum(0) - ref).max() < 2e-3 * np.abs(ref).max()
    # GELU epilogue vs torch
    a = torch.from_numpy(rng.standard_normal((48, 256)).astype(np.float32)).to(torch.bfloat16)
    w = torch.from_numpy((0.2 * rng.standard_normal((64, 256))).astype(np.float32)).to(torch.bfloat16)
    bias = T(rng.standard_normal(64).astype(np.float32))
    ref = torch.nn.functional.gelu(a.float() @ w.float().T + bias).numpy()
    out = nat.dbg_gemm(a, w, bias, epi=1, bm=66, bn=32, splitk=1).float().numpy()
    assert np.abs(out - ref).max() < 1e-2 * np.abs(ref).max()
    assert np.array_equal(nat.dbg_gemm(a, w, bias, epi=1, bm=66, bn=64, splitk=1).float().numpy(), out)


def test_emu_conv_halo(nat):
    """halo-reuse 3x3 conv (csrc/conv_halo.hip): plain, with fused GroupNorm+SiLU on the input, with residual, one and two
    channel chunks; against the oracle's conv2d / silu on the bf16-rounded operands."""
    from oracle.vae import conv2d, silu
    rng = np.random.default_rng(4)
    B, H, W, Cin, Cout = 1, 64, 64, 64, 128

    def bf(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16)
    x = bf(rng.standard_normal((B, H, W, Cin)).astype(np.float32))
    w = bf((0.05 * rng.standard_normal((Cout, 3, 3, Cin))).astype(np.float32))
    bias = T(rng.standard_normal(Cout).astype(np.float32))
    resid = bf(rng.standard_normal((B, H, W, Cout)).astype(np.float32))
    gn = T(np.stack([1.0 + 0.2 * rng.standard_normal((B, Cin)), 0.3 * rng.standard_normal((B, Cin))], -1).astype(np.float32))
    xf, wf = x.float().numpy(), np.transpose(w.float().numpy(), (0, 3, 1, 2))
    ref_plain = conv2d(xf, wf, bias.numpy())
    xn = silu(xf * gn.numpy()[:, None, None, :, 0] + gn.numpy()[:, None, None, :, 1])
    xn = bf(xn.astype(np.float32)).float().numpy()                     # the kernel rounds the activated input to bf16
    ref_gn = conv2d(xn, wf, bias.numpy()) + resid.float().numpy()
    th = 8
    out = nat.dbg_conv_halo(x, w, bias).float().numpy()
    assert np.abs(out - ref_plain).max() < 0.02 * np.abs(ref_plain).max()
    stats = torch.zeros((B, (H // th) * (W // 32), 32, 2), dtype=torch.float32)
    out = nat.dbg_conv_halo(x, w, bias, gn=gn, resid=resid, stats=stats).float().numpy()
    assert np.abs(out - ref_gn).max() < 0.02 * np.abs(ref_gn).max()
    # epilogue statistics: per (8 x 32 tile, group of Cout/32 channels) sum and sum of squares of the bf16 output
    t = out.reshape(B, H // th, th, W // 32, 32, 32, Cout // 32).astype(np.float64)
    want = np.stack([t.sum((2, 4, 6)), (t * t).sum((2, 4, 6))], -1).reshape(B, -1, 32, 2)
    assert np.abs(stats.numpy() - want).max() < 1e-3 * np.abs(want).max()
    # two channel chunks (the patch double buffer and the weight ring wrap) and two cout tiles
    x2 = bf(rng.standard_normal((1, 64, 32, 128)).astype(np.float32))
    w2 = bf((0.05 * rng.standard_normal((256, 3, 3, 128))).astype(np.float32))
    b2 = T(rng.standard_normal(256).astype(np.float32))
    ref2 = conv2d(x2.float().numpy(), np.transpose(w2.float().numpy(), (0, 3, 1, 2)), b2.numpy())
    out = nat.dbg_conv_halo(x2, w2, b2, persistent=False).float().numpy()
    assert np.abs(out - ref2).max() < 0.02 * np.abs(ref2).max()
    # 32 x 32 maps (one tile wide, four tall: what large batches run at the 32^2 level), fused GroupNorm + residual + statistics
    x3 = bf(rng.standard_normal((2, 32, 32, Cin)).astype(np.float32))
    r3 = bf(rng.standard_normal((2, 32, 32, Cout)).astype(np.float32))
    gn3 = T(np.stack([1.0 + 0.2 * rng.standard_normal((2, Cin)), 0.3 * rng.standard_normal((2, Cin))], -1).astype(np.float32))
    xn3 = bf(silu(x3.float().numpy() * gn3.numpy()[:, None, None, :, 0] + gn3.numpy()[:, None, None, :, 1]).astype(np.float32)).float().numpy()
    ref3 = conv2d(xn3, wf, bias.numpy()) + r3.float().numpy()
    st3 = torch.zeros((2, 4, 32, 2), dtype=torch.float32)
    out = nat.dbg_conv_halo(x3, w, bias, gn=gn3, resid=r3, stats=st3).float().numpy()
    assert np.abs(out - ref3).max() < 0.02 * np.abs(ref3).max()
    t3 = out.reshape(2, 4, 8, 1, 32, 32, Cout // 32).astype(np.float64)
    want3 = np.stack([t3.sum((2, 4, 6)), (t3 * t3).sum((2, 4, 6))], -1).reshape(2, -1, 32, 2)
    assert np.abs(st3.numpy() - want3).max() < 1e-3 * np.abs(want3).max()
    # four channel chunks, fused GroupNorm + residual: the four-slot LDS-DMA weight ring of the per-tile kernel turns by 9 units per
    # chunk, so every chunk starts in a different slot (unit u in slot u & 3), and the last chunk's requests stop three taps early
    x4 = bf(rng.standard_normal((1, 32, 32, 256)).astype(np.float32))
    w4 = bf((0.05 * rng.standard_normal((128, 3, 3, 256))).astype(np.float32))
    r4 = bf(rng.standard_normal((1, 32, 32, 128)).astype(np.float32))
    gn4 = T(np.stack([1.0 + 0.2 * rng.standard_normal((1, 256)), 0.3 * rng.standard_normal((1, 256))], -1).astype(np.float32))
    xn4 = bf(silu(x4.float().numpy() * gn4.numpy()[:, None, None, :, 0] + gn4.numpy()[:, None, None, :, 1]).astype(np.float32)).float().numpy()
    ref4 = conv2d(xn4, np.transpose(w4.float().numpy(), (0, 3, 1, 2)), bias.numpy()) + r4.float().numpy()
    out = nat.dbg_conv_halo(x4, w4, bias, gn=gn4, resid=r4).float().numpy()
    assert np.abs(out - ref4).max() < 0.02 * np.abs(ref4).max()
    # eight channel chunks (the ring turns through its four slots 18 times per tile), plain + residual
    x8 = bf(rng.standard_normal((1, 32, 32, 512)).astype(np.float32))
    w8 = bf((0.03 * rng.standard_normal((128, 3, 3, 512))).astype(np.float32))
    r8 = bf(rng.standard_normal((1, 32, 32, 128)).astype(np.float32))
    ref8 = conv2d(x8.float().numpy(), np.transpose(w8.float().numpy(), (0, 3, 1, 2)), bias.numpy()) + r8.float().numpy()
    out = nat.dbg_conv_halo(x8, w8, bias, resid=r8).float().numpy()
    assert np.abs(out - ref8).max() < 0.02 * np.abs(ref8).max()
    xs3 = bf(rng.standard_normal((1, 16, 16, Cin)).astype(np.float32))
    ref_up3 = conv2d(np.repeat(np.repeat(xs3.float().numpy(), 2, axis=1), 2, axis=2), wf, bias.numpy())
    out = nat.dbg_conv_halo(xs3, w, bias, ups=True).float().numpy()
    assert out.shape == (1, 32, 32, Cout) and np.abs(out - ref_up3).max() < 0.02 * np.abs(ref_up3).max()
    # Upsample.conv (layers.py:31-35): nearest 2x folded into the patch staging; 32x32 source -> 64x64 output
    xs = bf(rng.standard_normal((B, H // 2, W // 2, Cin)).astype(np.float32))
    xu = np.repeat(np.repeat(xs.float().numpy(), 2, axis=1), 2, axis=2)
    ref_up = conv2d(xu, wf, bias.numpy())
    for pers in (False, True):
        out = nat.dbg_conv_halo(xs, w, bias, ups=True, persistent=pers).float().numpy()
        assert out.shape == (B, H, W, Cout)
        assert np.abs(out - ref_up).max() < 0.02 * np.abs(ref_up).max(), pers


def test_emu_conv_halo_weight_dma_lands_late(nat, monkeypatch):
    """The per-tile halo conv's weight ring is filled by LDS-DMA (round 5): the same cases with every DMA landing at the LATEST moment its
    wavefront's counted wait allows (RQ_EMU_DMA=late; the default run lands them at issue, the worst case for a slot that is still being
    read).  A weight fragment read that is not behind the covering wait + barrier returns the slot's previous unit."""
    monkeypatch.setenv('RQ_EMU_DMA', 'late')
    test_emu_conv_halo(nat)


def test_emu_conv_halo_persistent(nat):
    """persistent form of the 8-row halo conv (the folded-upsample convs: a workgroup walks several tiles, staging the next
    tile's patch / weights during the last chunk of the current one, epilogue tile placed clear of the staged patch): one
    workgroup per XCD and the default count; two channel chunks, two cout tiles (the weight base switches between consecutive
    slots); must equal the per-tile kernel bit for bit.  (Persistent forms of the plain / fused convs are not compiled.)"""
    rng = np.random.default_rng(14)

    def bf(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16)
    for (B, Hs, Ws, Cin, Cout) in ((1, 32, 32, 128, 128), (2, 32, 16, 128, 256)):
        xs = bf(rng.standard_normal((B, Hs, Ws, Cin)).astype(np.float32))
        w = bf((0.05 * rng.standard_normal((Cout, 3, 3, Cin))).astype(np.float32))
        bias = T(rng.standard_normal(Cout).astype(np.float32))
        a = nat.dbg_conv_halo(xs, w, bias, ups=True, persistent=False)
        for wpx in (1, 0):
            b = nat.dbg_conv_halo(xs, w, bias, ups=True, persistent=True, wpx=wpx)
            assert torch.equal(a, b), (Cin, Cout, wpx)


def test_emu_conv_halo_upsample_subpixel(nat):
    """Round 5: Upsample.conv (layers.py:20-35) by sub-pixel decomposition -- for each output parity a 2 x 2 conv over the SOURCE image
    with the taps that share a source pixel summed (conv3x3_halo_pk_kernel<0, 2, 0>: 4 taps per output pixel instead of 9).  The
    pre-summed weights against a numpy sum of the taps; the conv against the oracle's conv2d of the nearest-upsampled input (image
    borders, two channel chunks, two cout tiles, several tiles per workgroup, one workgroup per XCD) and close to the 9-tap folded
    form; the epilogue statistics against sums over the output's 8 x 32-pixel lattices."""
    from oracle.vae import conv2d
    rng = np.random.default_rng(24)

    def bf(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16)
    for (B, Hs, Ws, Cin, Cout) in ((1, 16, 32, 64, 128), (2, 16, 32, 128, 256)):
        xs = bf(rng.standard_normal((B, Hs, Ws, Cin)).astype(np.float32))
        w = bf((0.05 * rng.standard_normal((Cout, 3, 3, Cin))).astype(np.float32))
        bias = T(rng.standard_normal(Cout).astype(np.float32))
        wf = w.float().numpy()
        # weights: class (py, px), tap (a, b) = the 3 x 3 taps that read source pixel (y + py - 1 + a, x + px - 1 + b)
        wsub = nat.dbg_ups_subpixel_weights(w).float().numpy()
        sets = {0: ([0], [1, 2]), 1: ([0, 1], [2])}
        for py in (0, 1):
            for px in (0, 1):
                for a in (0, 1):
                    for b in (0, 1):
                        want = sum(wf[:, ky, kx, :] for ky in sets[py][a] for kx in sets[px][b])
                        got = wsub[py * 2 + px, :, a, b, :]
                        assert np.abs(got - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6, (py, px, a, b)      # one bf16 rounding
        xu = np.repeat(np.repeat(xs.float().numpy(), 2, axis=1), 2, axis=2)
        ref = conv2d(xu, np.transpose(wf, (0, 3, 1, 2)), bias.numpy())
        H, W = 2 * Hs, 2 * Ws
        stats = torch.zeros((B, (H // 8) * (W // 32), 32, 2), dtype=torch.float32)
        out = nat.dbg_conv_halo(xs, w, bias, ups=True, subpixel=True, stats=stats)
        o = out.float().numpy()
        assert o.shape == (B, H, W, Cout)
        err = np.abs(o - ref).max() / np.abs(ref).max()
        folded = nat.dbg_conv_halo(xs, w, bias, ups=True).float().numpy()
        print(f'emu sub-pixel upsample conv {Cin}->{Cout} @ {Hs}x{Ws}: max err {err:.4f} of max vs fp32; vs the 9-tap folded form {np.abs(o - folded).max() / np.abs(ref).max():.4f}')
        assert err < 0.02
        assert np.abs(o - folded).max() < 0.02 * np.abs(ref).max()
        assert torch.equal(out, nat.dbg_conv_halo(xs, w, bias, ups=True, subpixel=True, wpx=1))          # one workgroup per XCD walks every tile
        # statistics: one partial per (source tile, parity class) = per 8 x 32 lattice of output pixels; their sum over an image is the
        # image's sum (what gn_params_kernel forms), and every partial is the sum over its own lattice
        tot = stats.numpy().astype(np.float64).sum(1)                                    # (B, 32, 2)
        t = o.reshape(B, H * W, 32, Cout // 32).astype(np.float64)
        want = np.stack([t.sum((1, 3)), (t * t).sum((1, 3))], -1)
        assert np.abs(tot - want).max() < 1e-3 * np.abs(want).max()
        lat = o[:, 0:16:2, 1:64:2].reshape(B, -1, 32, Cout // 32).astype(np.float64)     # source tile 0, class (py 0, px 1) = partial 1
        want1 = np.stack([lat.sum((1, 3)), (lat * lat).sum((1, 3))], -1)
        assert np.abs(stats.numpy()[:, 1] - want1).max() < 1e-3 * np.abs(want1).max()


def test_emu_conv_in_mfma(nat):
    """Encoder.conv_in as an MFMA kernel: NCHW fp32 image -> NHWC bf16, K = 27 padded to 32, image borders."""
    from oracle.vae import conv2d
    rng = np.random.default_rng(8)
    B, H, W = 2, 16, 32
    x = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    w = (0.2 * rng.standard_normal((128, 3, 3, 3))).astype(np.float32)         # (cout, ci, ky, kx) like nn.Conv2d
    bias = rng.standard_normal(128).astype(np.float32)
    bf = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()
    ref = conv2d(bf(np.transpose(x, (0, 2, 3, 1))), bf(w), bias)              # NHWC in, (cout, ci, ky, kx) weights
    w_k = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)))                  # (ky, kx, ci, cout)
    out = nat.dbg_conv_in(T(x), T(w_k), T(bias)).float().numpy()
    assert out.shape == (B, H, W, 128)
    assert np.abs(out - ref).max() < 1e-2 * np.abs(ref).max()


def test_emu_conv_out_mfma(nat, monkeypatch):
    """Decoder.conv_out as an MFMA halo kernel (csrc/conv_halo.hip): Cin -> 3, NCHW fp32 out, optional fused
    norm_out GroupNorm+SiLU; image borders, two channel planes, against the oracle's conv2d.  8 workgroups walk the 24
    tiles (three each, one range crossing an image boundary: the next tile's patch is requested under the current tile's MFMAs)."""
    from oracle.vae import conv2d, silu
    monkeypatch.setenv('RQAMD_CONV_OUT_WGS', '8')
    rng = np.random.default_rng(6)
    B, H, W, Cin, Cout = 3, 16, 64, 128, 3
    x = torch.from_numpy(rng.standard_normal((B, H, W, Cin)).astype(np.float32)).to(torch.bfloat16)
    w = (0.05 * rng.standard_normal((Cout, 3, 3, Cin))).astype(np.float32)
    bias = rng.standard_normal(Cout).astype(np.float32)
    gn = np.stack([1.0 + 0.2 * rng.standard_normal((B, Cin)), 0.3 * rng.standard_normal((B, Cin))], -1).astype(np.float32)
    w_bf = torch.from_numpy(w).to(torch.bfloat16).float().numpy()      # the kernel multiplies bf16 weights
    wf = np.transpose(w_bf, (0, 3, 1, 2))
    xf = x.float().numpy()
    ref = np.transpose(conv2d(xf, wf, bias), (0, 3, 1, 2))
    out = nat.dbg_conv_out(x, T(w), T(bias)).numpy()
    assert out.shape == (B, Cout, H, W)
    assert np.abs(out - ref).max() < 2e-3 * np.abs(ref).max() + 1e-4
    xn = silu(xf * gn[:, None, None, :, 0] + gn[:, None, None, :, 1])
    xn = torch.from_numpy(xn.astype(np.float32)).to(torch.bfloat16).float().numpy()
    ref = np.transpose(conv2d(xn, wf, bias), (0, 3, 1, 2))
    out = nat.dbg_conv_out(x, T(w), T(bias), gn=T(gn)).numpy()
    assert np.abs(out - ref).max() < 0.02 * np.abs(ref).max()
