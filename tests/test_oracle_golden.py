"""Pin the numpy oracle against fixtures produced by the reference itself
(tests/golden/make_golden.py, run in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest

import oracle
from oracle import configs as C

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _rq_inputs(g):
    rng = np.random.default_rng(int(g['seed']))
    K, D, N = int(g['K']), int(g['D']), int(g['N'])
    cb = rng.standard_normal((K, D), dtype=np.float32)
    x = rng.standard_normal((N // 64, 8, 8, D), dtype=np.float32)
    return x, cb


def test_rq_small(golden):
    g = golden('rq_small.npz')
    x, cb = g['x'], g['codebook']
    quant_list, codes = oracle.rq_quantize(x, [cb] * 4)
    assert np.array_equal(codes, g['codes'])                       # bit-exact indices
    assert np.array_equal(np.stack(quant_list), g['quant_list'])   # bit-exact fp32 cumulative quants
    assert np.array_equal(oracle.rq_embed_code(codes, [cb] * 4), g['quant_list'][-1])
    assert np.array_equal(oracle.rq_embed_code_with_depth(codes.reshape(-1, 64, 4), [cb] * 4), g['embed_with_depth'])
    gaps, c64 = oracle.rq_quantize_margins(x, [cb] * 4)
    assert np.array_equal(c64, g['codes'])
    np.testing.assert_allclose(gaps, g['gaps'], rtol=1e-4, atol=1e-5)


def test_rq_full_size(golden):
    g = golden('rq_full.npz')
    x, cb = _rq_inputs(g)
    quant_list, codes = oracle.rq_quantize(x, [cb] * 4)
    assert np.array_equal(codes, g['codes'])
    np.testing.assert_allclose(quant_list[-1].astype(np.float64).sum(-1), g['quant_last_sum'], rtol=0, atol=1e-3)


def test_sampler_filters(golden):
    g = golden('sampler.npz')
    tie_free = [0, 2, 4, 5, 6, 7]          # rows 1 (rounded logits) and 3 (uniform) have exact ties
    for i, (t, k, p) in enumerate(g['cases']):
        ref = g[f'probs_{i}']
        o = oracle.filtered_probs(g['logits'], t, None if k < 0 else int(k), None if p < 0 else float(p))
        assert 0.5 * np.abs(o - ref).sum(-1)[tie_free].max() < 2e-6
        # tie rows: which of several equal probabilities survives top-p is sort-order defined
        # (SURVEY §3.3); the multiset of kept probabilities is not.
        assert np.abs(np.sort(o, -1) - np.sort(ref, -1)).sum(-1).max() < 5e-5
        assert np.array_equal((o > 0).sum(-1), (ref > 0).sum(-1)) or p == 1.0


def test_vae_tiny(golden):
    g = golden('vae_tiny.npz')
    hps, dd = C.VAE_TINY
    ov = oracle.RQVAEOracle(hps, dd, oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['seed'])))
    np.testing.assert_allclose(ov.encode(g['x']), g['z_e'], rtol=0, atol=2e-5)
    assert np.array_equal(ov.get_codes(g['x']), g['codes'])
    np.testing.assert_allclose(ov.decode_code(g['codes']), g['decode_code'], rtol=0, atol=2e-5)
    out, loss, codes = ov.forward(g['x'])
    np.testing.assert_allclose(out, g['forward_out'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(loss, g['loss'], rtol=1e-5)


def test_vae_imagenet_decode(golden):
    g = golden('vae_imagenet.npz')
    hps, dd = C.VAE_IMAGENET
    ov = oracle.RQVAEOracle(hps, dd, oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['seed'])))
    dec = ov.decode_code(g['codes'])
    # fixture is stored as fp16: tolerance = fp16 rounding of |x| <= 4.3
    np.testing.assert_allclose(dec, g['decode_code'].astype(np.float32), rtol=0, atol=3e-3)


def test_rqt_tiny(golden):
    g = golden('rqt_tiny.npz')
    cfg = C.RQT_TINY
    hps, dd = C.VAE_TINY
    cb = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['vae_seed']))['quantizer.codebooks.0.weight'][:-1]
    orc = oracle.RQTransformerOracle(cfg, oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed'])))
    codes, cond = g['codes'].astype(np.int64), g['cond'].astype(np.int64)
    np.testing.assert_allclose(orc.forward(codes, [cb] * 4, cond), g['logits'], rtol=0, atol=1e-5)
    H, W, D = cfg['block_size']
    orc.init_cache()
    for h in range(H):
        for w in range(W):
            for d in range(D):
                lg = orc.cached_forward(codes[:, :h + 1], [cb] * 4, cond, (h, w, d))
                np.testing.assert_allclose(lg, g['logits'][:, h, w, d], rtol=0, atol=1e-5)
    # start_loc prefill path
    orc.init_cache()
    lg = orc.cached_forward(codes[:, :2], [cb] * 4, cond, (1, 2, 0))
    np.testing.assert_allclose(lg, g['logits'][:, 1, 2, 0], rtol=0, atol=1e-5)


def test_rqt_tiny_text_conditioned(golden):
    """block_size_cond = 4: conditioning prefix + prefill of the first cached step (transformers.py:235-239)."""
    g = golden('rqt_tiny_txt.npz')
    cfg = C.RQT_TINY_TXT
    hps, dd = C.VAE_TINY
    cb = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), int(g['vae_seed']))['quantizer.codebooks.0.weight'][:-1]
    orc = oracle.RQTransformerOracle(cfg, oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed'])))
    codes, cond = g['codes'].astype(np.int64), g['cond'].astype(np.int64)
    assert cond.shape == (2, 4)
    np.testing.assert_allclose(orc.forward(codes, [cb] * 4, cond), g['logits'], rtol=0, atol=1e-5)
    orc.init_cache()
    for h in range(4):
        for w in range(4):
            lg = orc.cached_forward(codes[:, :h + 1], [cb] * 4, cond, (h, w, 0))
            np.testing.assert_allclose(lg, g['logits'][:, h, w, 0], rtol=0, atol=1e-5)


@pytest.mark.parametrize('tag', ['xwide', 'txt32', 'txt64'])
def test_oracle_vs_reference_real_widths(golden, tag):
    """E=2560/40 heads and E=1280/20 heads with 32 / 64 text tokens (few layers): the oracle against the reference's
    own logits (seq logits at the stored positions; cond_classifier logits for the text shapes)."""
    g = golden(f'rqt_{tag}.npz')
    cfg = {'xwide': C.RQT_XWIDE, 'txt32': C.RQT_TXT32, 'txt64': C.RQT_TXT64}[tag]
    cb = np.random.default_rng(int(g['cb_seed'])).standard_normal((cfg['vocab_size'], 256), dtype=np.float32)
    orc = oracle.RQTransformerOracle(cfg, oracle.make_params(oracle.rqt_param_shapes(cfg), int(g['seed'])))
    out = orc.forward(g['codes'].astype(np.int64), [cb] * 4, g['cond'].astype(np.int64), return_cond_logits=True)
    seq = out[0] if isinstance(out, tuple) else out
    got = np.stack([seq[:, h, w] for h, w in g['pos']], 1)
    assert np.abs(got - g['logits'].astype(np.float32)).max() < 2e-3          # fp16 storage of |logits| <= 3.2
    if isinstance(out, tuple):
        assert np.abs(out[1][:, g['cond_pos']] - g['cond_logits'].astype(np.float32)).max() < 2e-3


def test_rq_ema_update(golden):
    """train-mode quantiser (EMA codebook update + dead-code restart, quantizations.py:80-142,237-271) vs the reference's outputs"""
    g = golden('rq_ema.npz')
    K, Dm, seed, decay = int(g['K']), int(g['D']), int(g['seed']), float(g['decay'])
    rng = np.random.default_rng(seed)
    cb = rng.standard_normal((K, Dm), dtype=np.float32)
    cs0 = rng.uniform(0.0, 3.0, K).astype(np.float32)
    xs = [rng.standard_normal((6, 8, 8, Dm), dtype=np.float32), rng.standard_normal((2, 8, 8, Dm), dtype=np.float32)]
    x_many = rng.standard_normal((384, Dm), dtype=np.float32)
    x_few = rng.standard_normal((128, Dm), dtype=np.float32)
    w, cs, ee = cb, cs0, cb * cs0[:, None]
    for b, x in enumerate(xs):
        ql, codes, w, cs, ee = oracle.rq_quantize_train(x, w, cs, ee, 4, decay, 1e-5, None)
        assert np.array_equal(codes, g[f'codes{b}'])
        np.testing.assert_allclose(ql[-1], g[f'quant_last{b}'], rtol=0, atol=2e-6)
        for got, key in ((w, 'weight'), (cs, 'cs'), (ee, 'ee')):
            np.testing.assert_allclose(got, g[f'{key}{b}'], rtol=1e-5, atol=1e-6)
    for tag, xv, sd in (('many', x_many, seed + 1), ('few', x_few, seed + 2)):
        rv = oracle.ema_restart_candidates(xv, K, np.random.default_rng(sd))
        quant, code, w1, cs1, ee1 = oracle.vq_ema_step(cb, cs0, cb * cs0[:, None], xv, decay, 1e-5, rv)
        assert np.array_equal(code, g[f'{tag}_codes']) and np.array_equal(quant, g[f'{tag}_embeds'])
        assert np.array_equal(cs1 == 1, g[f'{tag}_cs'] == 1) and (cs1 == 1).sum() > 50        # the same codes restarted
        for got, key in ((w1, 'weight'), (cs1, 'cs'), (ee1, 'ee')):
            np.testing.assert_allclose(got, g[f'{tag}_{key}'], rtol=1e-5, atol=1e-6)


def test_param_counts():
    """README.md:38-47 of the reference: structural known answers."""
    with open(os.path.join(GOLDEN, 'param_counts.json')) as f:
        ref = json.load(f)
    for name, want_m in C.PARAM_COUNTS_M.items():
        n = sum(int(np.prod(s)) for s in oracle.rqt_param_shapes(getattr(C, name)).values())
        assert n == ref[name][0]
        assert abs(n / 1e6 - want_m) < 0.06
    hps, dd = C.VAE_IMAGENET
    shapes = oracle.rqvae_param_shapes(hps, dd)
    n = sum(int(np.prod(s)) for k, s in shapes.items()
            if 'ema' not in k and not any(k.startswith(f'quantizer.codebooks.{i}.') for i in (1, 2, 3)))
    assert abs(n / 1e6 - 104.4) < 0.06


def test_torch_backend_agrees_with_numpy_forms():
    """oracle/backend.py: the torch-CPU forms of the heavy primitives (used only by bench.py's cpu_baseline leg) compute what
    the pinned numpy forms compute."""
    from oracle import backend, vae as ovae, transformer as otr
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 8, 8, 64)).astype(np.float32)
    w = (0.1 * rng.standard_normal((32, 64, 3, 3))).astype(np.float32)
    b = rng.standard_normal(32).astype(np.float32)
    g, be = rng.standard_normal(64).astype(np.float32), rng.standard_normal(64).astype(np.float32)
    lw = rng.standard_normal((48, 64)).astype(np.float32)
    want = [ovae.conv2d(x, w, b), ovae.conv2d(x, w, b, stride=2, pad=(0, 1, 0, 1)), ovae.group_norm(x, g, be), ovae.silu(x),
            otr.linear(x, lw, b[:1].repeat(48)), otr.gelu(x), otr.gelu(x, 'v2')]
    backend.use_torch(True)
    try:
        got = [ovae.conv2d(x, w, b), ovae.conv2d(x, w, b, stride=2, pad=(0, 1, 0, 1)), ovae.group_norm(x, g, be), ovae.silu(x),
               otr.linear(x, lw, b[:1].repeat(48)), otr.gelu(x), otr.gelu(x, 'v2')]
    finally:
        backend.use_torch(False)
    for a, c in zip(want, got):
        assert a.shape == c.shape
        assert np.abs(a - c).max() <= 2e-5 * max(1.0, np.abs(a).max())
