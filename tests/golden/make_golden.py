#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF on CPU.

Run in the build container only (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_golden.py

The reference (kakaobrain/rq-vae-transformer) is imported from /root/reference behind a
stub for the one missing import (omegaconf, rqvae/models/rqtransformer/configs.py:18; the
model classes never call it).  Weights come from oracle.weights.make_params (seeded, keyed by
state_dict name) and are loaded with load_state_dict(strict=True), so the fixtures also pin the
reference's key/shape tables.  Each fixture stores only seeds + reference outputs; inputs and
weights are regenerated from the seeds by the tests.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

_m = types.ModuleType('omegaconf')
_m.OmegaConf = type('OmegaConf', (), {})
_m.MISSING = '???'
_m.DictConfig = dict
sys.modules['omegaconf'] = _m
sys.path.insert(0, '/root/reference')

import torch  # noqa: E402
from rqvae.models.rqvae import RQVAE  # noqa: E402  (reference)
from rqvae.models.rqtransformer import RQTransformer  # noqa: E402  (reference)
from rqvae.utils.utils import top_k_logits, top_p_probs  # noqa: E402  (reference)
import torch.nn.functional as F  # noqa: E402

import oracle  # noqa: E402
from oracle import configs as C  # noqa: E402

torch.set_grad_enabled(False)


class Cfg(dict):
    """attribute-dict with .copy() -- all RQTransformer.__init__ needs (transformers.py:39-52)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def copy(self):
        return to_cfg(json.loads(json.dumps(self)))


def to_cfg(d):
    return Cfg({k: to_cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


def ref_rqvae(hps, dd, seed):
    m = RQVAE(**hps, ddconfig=dd, checkpointing=False).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    mine = oracle.rqvae_param_shapes(hps, dd)
    assert shapes == {k: tuple(v) for k, v in mine.items()}, 'rqvae key/shape table mismatch'
    params = oracle.make_params(mine, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m, params


def ref_rqt(cfg, seed):
    m = RQTransformer(to_cfg(cfg)).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    mine = oracle.rqt_param_shapes(cfg)
    assert shapes == {k: tuple(v) for k, v in mine.items()}, 'rqt key/shape table mismatch'
    params = oracle.make_params(mine, seed, cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m, params


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f'  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB')


def rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


# ------------------------------------------------------------------ 1. residual quantiser
def gen_rq():
    from rqvae.models.rqvae.quantizations import RQBottleneck
    for tag, K, Dm, N, seed in (('small', 500, 64, 192, 11), ('full', 16384, 256, 1024, 12)):
        rng = np.random.default_rng(seed)
        cb = rng.standard_normal((K, Dm), dtype=np.float32)
        x = rng.standard_normal((N // 64, 8, 8, Dm), dtype=np.float32)
        if tag == 'small':      # half the vectors near codewords (realistic margins)
            x.reshape(-1, Dm)[::2] = cb[rng.integers(0, K, N // 2)] * 2.5 + 0.1 * x.reshape(-1, Dm)[::2]
        rq = RQBottleneck([8, 8, Dm], [8, 8, 4], K, shared_codebook=True).eval()
        w = np.concatenate([cb, np.zeros((1, Dm), np.float32)])
        rq.codebooks[0].weight.data.copy_(torch.from_numpy(w))
        quant_list, codes = rq.quantize(torch.from_numpy(x))
        emb = rq.embed_code(codes)
        embd, _ = rq.embed_code_with_depth(codes.reshape(N // 64, 64, 4))
        assert torch.equal(emb, quant_list[-1])
        gaps, codes64 = oracle.rq_quantize_margins(x, [cb] * 4)
        oq, oc = oracle.rq_quantize(x, [cb] * 4)
        print(f'  rq[{tag}] oracle==ref codes: {(oc == codes.numpy()).mean():.6f}, fp64==ref: '
              f'{(codes64 == codes.numpy()).mean():.6f}, min gap {gaps.min():.2e}')
        arrs = dict(seed=seed, K=K, D=Dm, N=N, codes=codes.numpy().astype(np.int32), gaps=gaps.astype(np.float32),
                    quant_last_sum=quant_list[-1].numpy().astype(np.float64).sum(-1).astype(np.float32))
        if tag == 'small':
            arrs.update(x=x, codebook=cb, quant_list=np.stack([q.numpy() for q in quant_list]),
                        embed_with_depth=embd.numpy())
        save(f'rq_{tag}.npz', **arrs)
        if tag == 'small':      # soft codes: reference get_soft_codes on the first image, temp 2.0
            soft, scode = rq.get_soft_codes(torch.from_numpy(x[:1]), temp=2.0, stochastic=False)
            osoft, ocode = oracle.rq_soft_codes(x[:1], [cb] * 4, temp=2.0)
            print(f'  rq[soft] oracle vs ref: soft codes {np.abs(osoft - soft.numpy()).max():.2e}, codes equal {(ocode == scode.numpy()).all()}')
            assert torch.equal(scode, codes[:1])
            save('rq_soft.npz', seed=seed, temp=np.float32(2.0), soft=soft.numpy().astype(np.float32), codes=scode.numpy().astype(np.int32))


# ------------------------------------------------------------------ 2. sampler filters
def gen_sampler():
    rng = np.random.default_rng(21)
    V = 1024
    logits = (3.0 * rng.standard_normal((8, V))).astype(np.float32)
    logits[1, :] = np.round(logits[1, :])            # many exact ties (incl. at the k-th value)
    logits[2, 5] = np.nan                            # NaN scrub path (utils.py:103-105)
    logits[3, :] = 0.0                               # uniform row: every prob tied
    logits[4, 7] = 40.0                              # one dominant token
    cases = [(1.0, None, None), (1.0, V, 1.0), (1.0, 100, None), (0.8, 100, 0.95), (1.0, None, 0.5),
             (1.3, 1, None), (1.0, 1024, 0.95), (0.5, 17, 0.3)]
    out = {}
    for i, (t, k, p) in enumerate(cases):
        x = torch.from_numpy(logits).to(torch.float32) / t
        if k is not None:
            x = top_k_logits(x, k)
        x[torch.isnan(x)] = -float('inf')
        probs = F.softmax(x, dim=-1)
        if p is not None:
            probs = top_p_probs(probs, p)
        out[f'probs_{i}'] = probs.numpy()
        o = oracle.filtered_probs(logits, t, k, p)
        tv = 0.5 * np.abs(o - probs.numpy()).sum(-1)
        srt = np.abs(np.sort(o, -1) - np.sort(probs.numpy(), -1)).sum(-1)   # tie-order invariant
        print(f'  sampler case {i} (T={t},k={k},p={p}): oracle TV to ref, tie-free rows {tv[[0, 2, 4, 5, 6, 7]].max():.2e}; '
              f'sorted-multiset L1, all rows {srt.max():.2e}')
    save('sampler.npz', logits=logits, cases=np.array([[t, -1 if k is None else k, -1 if p is None else p]
                                                       for t, k, p in cases], np.float64), **out)


# ------------------------------------------------------------------ 3. tiny RQ-VAE + full-size decode/encode
def gen_vae():
    hps, dd = C.VAE_TINY
    m, params = ref_rqvae(hps, dd, seed=31)
    rng = np.random.default_rng(32)
    x = np.clip(rng.standard_normal((2, 3, 16, 16), dtype=np.float32), -1, 1)
    z_e = m.encode(torch.from_numpy(x))
    out, loss, codes = m(torch.from_numpy(x))
    dec = m.decode_code(codes)
    ov = oracle.RQVAEOracle(hps, dd, params)
    print(f'  vae[tiny] oracle vs ref: encode {rel(ov.encode(x), z_e.numpy()):.2e}, '
          f'decode_code {rel(ov.decode_code(codes.numpy()), dec.numpy()):.2e}, '
          f'codes equal {(ov.get_codes(x) == codes.numpy()).mean():.4f}')
    save('vae_tiny.npz', seed=31, x=x, z_e=z_e.numpy(), codes=codes.numpy().astype(np.int32), decode_code=dec.numpy(),
         forward_out=out.numpy(), loss=np.float32(loss.item()))

    # ddconfig.resamp_with_conv = False (round 6): bare nearest upsample / average pool
    hps, dd = C.VAE_TINY_NORESAMP
    m, params = ref_rqvae(hps, dd, seed=36)
    assert not any('sample.conv' in k for k in params)
    x = np.clip(np.random.default_rng(37).standard_normal((2, 3, 16, 16), dtype=np.float32), -1, 1)
    z_e = m.encode(torch.from_numpy(x))
    out, loss, codes = m(torch.from_numpy(x))
    dec = m.decode_code(codes)
    ov = oracle.RQVAEOracle(hps, dd, params)
    print(f'  vae[tiny, resamp_with_conv=False] oracle vs ref: encode {rel(ov.encode(x), z_e.numpy()):.2e}, '
          f'decode_code {rel(ov.decode_code(codes.numpy()), dec.numpy()):.2e}')
    save('vae_tiny_noresamp.npz', seed=36, x=x, z_e=z_e.numpy(), codes=codes.numpy().astype(np.int32), decode_code=dec.numpy())

    for tag, (hps, dd) in (('imagenet', C.VAE_IMAGENET), ('ffhq', C.VAE_FFHQ)):
        m, params = ref_rqvae(hps, dd, seed=33)
        rng = np.random.default_rng(34)
        codes = rng.integers(0, hps['n_embed'], (1, 8, 8, 4))
        dec = m.decode_code(torch.from_numpy(codes)).numpy()
        x = np.clip(rng.standard_normal((1, 3, 256, 256), dtype=np.float32), -1, 1)
        z_e = m.encode(torch.from_numpy(x)).numpy()
        ecodes = m.get_codes(torch.from_numpy(x)).numpy()
        ov = oracle.RQVAEOracle(hps, dd, params)
        print(f'  vae[{tag}] oracle vs ref: decode_code {rel(ov.decode_code(codes), dec):.2e}, '
              f'encode {rel(ov.encode(x), z_e):.2e}; |dec| max {np.abs(dec).max():.3f} std {dec.std():.3f}')
        save(f'vae_{tag}.npz', seed=33, data_seed=34, codes=codes.astype(np.int32), decode_code=dec.astype(np.float16),
             z_e=z_e, enc_codes=ecodes.astype(np.int32), dec_absmax=np.float32(np.abs(dec).max()))


def gen_vae_batch():
    """Round 4: the released ImageNet RQ-VAE shape on a BATCH -- decode_code of 4 code maps (stored fp16: the
    tests report PSNR and the post-clamp uint8 difference, main_sampling_fid.py:223-225) and encode / get_codes of 8 images."""
    hps, dd = C.VAE_IMAGENET
    m, params = ref_rqvae(hps, dd, seed=33)
    rng = np.random.default_rng(35)
    codes = rng.integers(0, hps['n_embed'], (4, 8, 8, 4))
    dec = m.decode_code(torch.from_numpy(codes)).numpy()
    x = np.clip(rng.standard_normal((8, 3, 256, 256), dtype=np.float32), -1, 1)
    xt = torch.from_numpy(x)
    z_e = torch.cat([m.encode(xt[i:i + 2]) for i in range(0, 8, 2)]).numpy()
    ecodes = torch.cat([m.get_codes(xt[i:i + 2]) for i in range(0, 8, 2)]).numpy()
    print(f'  vae[imagenet batch] |dec| max {np.abs(dec).max():.3f} std {dec.std():.3f}; |z_e| max {np.abs(z_e).max():.3f}')
    save('vae_imagenet_batch.npz', seed=33, data_seed=35, codes=codes.astype(np.int32), decode_code=dec.astype(np.float16),
         z_e=z_e.astype(np.float32), enc_codes=ecodes.astype(np.int32))


# ------------------------------------------------------------------ 4. tiny RQ-Transformer
def gen_rqt():
    hps, dd = C.VAE_TINY
    vae, vparams = ref_rqvae(hps, dd, seed=31)
    cb = vparams['quantizer.codebooks.0.weight'][:-1]
    for tag, cfg, B in (('tiny', C.RQT_TINY, 3), ('tiny_txt', C.RQT_TINY_TXT, 2)):
        m, params = ref_rqt(cfg, seed=41)
        H, W, D = cfg['block_size']
        rng = np.random.default_rng(42)
        codes = rng.integers(0, cfg['vocab_size'], (B, H, W, D))
        cond = rng.integers(0, cfg['vocab_size_cond'], (B, max(cfg['block_size_cond'], 1)))
        tc, tcond = torch.from_numpy(codes), torch.from_numpy(cond)
        logits = m(tc, vae, cond=tcond)
        logits = (logits[0] if isinstance(logits, tuple) else logits).numpy()   # (seq_logits, cond_logits) when cond_len > 1
        # cached path == uncached path (the reference's built-in self check, transformers.py:352-356)
        m.init_cache()
        cl = np.zeros_like(logits)
        for h in range(H):
            for w in range(W):
                for d in range(D):
                    cl[:, h, w, d] = m.cached_forward(tc[:, :h + 1], vae, cond=tcond, sample_loc=(h, w, d)).numpy()
        m.init_cache()
        print(f'  rqt[{tag}] ref cached vs uncached logits: {np.abs(cl - logits).max():.2e}')
        orc = oracle.RQTransformerOracle(cfg, params)
        ol = orc.forward(codes, [cb] * D, cond)
        print(f'  rqt[{tag}] oracle forward vs ref: {np.abs(ol - logits).max():.2e} (|logits| max {np.abs(logits).max():.2f})')
        orc.init_cache()
        oc = np.stack([orc.cached_forward(codes[:, :h + 1], [cb] * D, cond, (h, w, d))
                       for h in range(H) for w in range(W) for d in range(D)], 1).reshape(logits.shape)
        print(f'  rqt[{tag}] oracle cached vs ref: {np.abs(oc - logits).max():.2e}')
        # start_loc > 0 prefill path (transformers.py:235-239)
        m.init_cache()
        pl = m.cached_forward(tc[:, :2], vae, cond=tcond, sample_loc=(1, 2, 0)).numpy()
        m.init_cache()
        print(f'  rqt[{tag}] ref prefill(start_loc=(1,2)) vs teacher-forced: {np.abs(pl - logits[:, 1, 2, 0]).max():.2e}')
        save(f'rqt_{tag}.npz', seed=41, vae_seed=31, codes=codes.astype(np.int32), cond=cond.astype(np.int32),
             logits=logits)


# ------------------------------------------------------------------ 4b. statistics of the reference's OWN sample()
from sample_stats import SAMPLE_STATS_N, SAMPLE_STATS_COARSE, sample_stats_inputs, sample_stats_counts  # noqa: E402  (tests/golden/sample_stats.py)


def gen_rqt_sample_stats():
    """RQTransformer.sample of the REFERENCE (transformers.py:294-369: cached_forward + sample_from_logits + torch.multinomial), tiny
    4 x 4 x 4 model, 20 000 images per case, full softmax (no top-k / top-p: a filter's boundary moves with bf16 rounding)."""
    hps, dd = C.VAE_TINY
    vae, _ = ref_rqvae(hps, dd, seed=31)
    cfg = C.RQT_TINY
    m, _ = ref_rqt(cfg, seed=41)
    out = {}
    for ci, case in enumerate(('cond', 'nocond', 'start')):
        part, cond, start = sample_stats_inputs(cfg, case)
        torch.manual_seed(9000 + ci)
        xs = m.sample(torch.from_numpy(part), vae, cond=None if cond is None else torch.from_numpy(cond), start_loc=start,
                      temperature=1.0, top_k=None, top_p=None, is_tqdm=False).numpy()
        marg, p_sp, p_dp = sample_stats_counts(xs, cfg, start)
        out[f'marg_{case}'], out[f'pair_spatial_{case}'], out[f'pair_depth_{case}'] = marg, p_sp, p_dp
        # a second, independent reference run of the same case: its chi^2 against the first is what "agreement" looks like
        torch.manual_seed(9100 + ci)
        xs2 = m.sample(torch.from_numpy(part), vae, cond=None if cond is None else torch.from_numpy(cond), start_loc=start,
                       temperature=1.0, top_k=None, top_p=None, is_tqdm=False).numpy()
        marg2, _, _ = sample_stats_counts(xs2, cfg, start)
        a, b = marg[0, 0].astype(np.float64), marg2[0, 0].astype(np.float64)
        keep = (a + b) >= 10
        print(f'  sample stats [{case}]: first position, depth 0: {int((marg[0, 0] > 0).sum())} of {cfg["vocab_size"]} codes drawn, '
              f'most frequent {marg[0, 0].max()} / {SAMPLE_STATS_N}; reference vs reference chi^2 {((a - b)[keep] ** 2 / (a + b)[keep]).sum():.1f} '
              f'on {int(keep.sum()) - 1} dof')
    save('rqt_tiny_sample_stats.npz', n=SAMPLE_STATS_N, seed=41, vae_seed=31, coarse=SAMPLE_STATS_COARSE, **out)


# ------------------------------------------------------------------ 5. RQ-Transformer at the benchmarked / released shapes
class CodebookAux:
    """minimal model_aux: the reference only calls get_code_emb_with_depth on it (transformers.py:109-111)"""

    def __init__(self, cb):
        self.cb = torch.from_numpy(cb)

    def get_code_emb_with_depth(self, xs):
        return F.embedding(xs, self.cb), None


BIG_POS = [(0, 0), (0, 1), (3, 5), (7, 7)]          # spatial positions whose logits are stored (all depths)


def big_codebook(K, seed):
    return np.random.default_rng(seed).standard_normal((K, 256), dtype=np.float32)


def gen_rqt_big(only=None):
    """Reference logits (teacher-forced forward(), fp32, CPU) for the configurations bench.py times and the
    other released widths: full 1.4B (42+6 layers), full FFHQ-355M (24+4, V=2048, unconditional), E=2560/40 heads
    (2+1 layers), and the text-to-image width E=1280/20 heads with 32 and 64 conditioning tokens (3+2 layers; also
    the cond_classifier logits).  Stored: fp16 logits at BIG_POS x all depths; inputs/weights regenerate from seeds."""
    cases = [('in1400m', C.RQT_IN_1400M, 2, 61), ('ffhq355m', C.RQT_FFHQ_355M, 2, 62), ('xwide', C.RQT_XWIDE, 2, 63),
             ('txt32', C.RQT_TXT32, 2, 64), ('txt64', C.RQT_TXT64, 2, 65),
             # round 4: BASELINE configs[3] / configs[4] at FULL depth (42 + 6 layers, E 2560); 15 GB of fp32
             # weights each on the CPU -- generate one at a time: `make_golden.py rqt_big:in3800m` / `rqt_big:txt3900m`
             ('in3800m', C.RQT_IN_3800M, 2, 66), ('txt3900m', C.RQT_TXT_3900M, 2, 67)]
    for tag, cfg, B, seed in cases:
        if only and tag not in only:
            continue
        m, params = ref_rqt(cfg, seed=seed)
        H, W, D = cfg['block_size']
        V = cfg['vocab_size']
        cb = big_codebook(V, seed + 100)
        rng = np.random.default_rng(seed + 200)
        codes = rng.integers(0, V, (B, H, W, D))
        cond = rng.integers(0, max(cfg['vocab_size_cond'], 1), (B, max(cfg['block_size_cond'], 1)))
        out = m(torch.from_numpy(codes), CodebookAux(cb), cond=torch.from_numpy(cond))
        extra = {}
        if isinstance(out, tuple):
            out, cond_logits = out
            cl = cond_logits.numpy()                                           # (B, cond_len-1, vocab_cond)
            cpos = sorted({0, 1, cl.shape[1] // 2, cl.shape[1] - 1})
            extra['cond_pos'] = np.array(cpos, np.int32)
            extra['cond_logits'] = cl[:, cpos].astype(np.float16)
        logits = out.numpy()
        sel = np.stack([logits[:, h, w] for h, w in BIG_POS], 1)          # (B, npos, D, V)
        print(f'  rqt[{tag}] |logits| max {np.abs(logits).max():.3f} std {logits.std():.3f}')
        if cfg['body']['n_layer'] + cfg['head']['n_layer'] <= 6:
            ol = oracle.RQTransformerOracle(cfg, params).forward(codes, [cb] * D, cond, return_cond_logits=True)
            if isinstance(ol, tuple):
                print(f'  rqt[{tag}] oracle cond_logits vs ref: {np.abs(ol[1] - cond_logits.numpy()).max():.2e}')
                ol = ol[0]
            print(f'  rqt[{tag}] oracle forward vs ref: {np.abs(ol - logits).max():.2e}')
        save(f'rqt_{tag}.npz', seed=seed, cb_seed=seed + 100, codes=codes.astype(np.int32), cond=cond.astype(np.int32),
             pos=np.array(BIG_POS, np.int32), logits=sel.astype(np.float16), logits_absmax=np.float32(np.abs(logits).max()),
             logits_std=np.float32(logits.std()), **extra)
        del m, params


def gen_rqt_variants():
    """Stage-2 flag variants that no released config uses (TupleEmbedding / BatchLinear / LogitMask, primitives.py:25-165;
    cumsum_depth_ctx off; shared learned head embedding; round 5: bias-free attention / MLP layers, configs.py:21-40): reference
    forward() logits on the tiny shape."""
    hps, dd = C.VAE_TINY
    vae, vparams = ref_rqvae(hps, dd, seed=31)
    cb = vparams['quantizer.codebooks.0.weight'][:-1]
    for tag, cfg in (('tuple', C.RQT_TINY_TUPLE), ('nocumsum', C.RQT_TINY_NOCUMSUM), ('mixed', C.RQT_TINY_MIXED), ('nobias', C.RQT_TINY_NOBIAS),
                     ('gelumix', C.RQT_TINY_GELUMIX), ('heads', C.RQT_TINY_HEADS), ('txtheads', C.RQT_TINY_TXT_HEADS)):
        m, params = ref_rqt(cfg, seed=47)
        H, W, D = cfg['block_size']
        vs = cfg['vocab_size'] if isinstance(cfg['vocab_size'], list) else [cfg['vocab_size']] * D
        rng = np.random.default_rng(48)
        codes = np.stack([rng.integers(0, v, (2, H, W)) for v in vs], -1)
        cond = rng.integers(0, cfg['vocab_size_cond'], (2, max(cfg['block_size_cond'], 1)))
        logits = m(torch.from_numpy(codes), vae, cond=torch.from_numpy(cond))
        ol = oracle.RQTransformerOracle(cfg, params).forward(codes, [cb] * D, cond)
        if isinstance(logits, tuple):         # text-conditioned: (seq_logits, cond_logits), transformers.py:185-186
            logits, ol = logits[0], (ol[0] if isinstance(ol, tuple) else ol)
        logits = logits.numpy()
        fin = np.isfinite(logits)
        assert np.array_equal(fin, np.isfinite(ol))
        print(f'  rqt[{tag}] oracle forward vs ref: {np.abs(ol[fin] - logits[fin]).max():.2e}; -inf entries {int((~fin).sum())}')
        save(f'rqt_var_{tag}.npz', seed=47, vae_seed=31, codes=codes.astype(np.int32), cond=cond.astype(np.int32), logits=logits)


def gen_ema():
    """Train-mode quantiser of the reference:
      * RQBottleneck.quantize, depth 4, ONE shared EMA codebook updated after every depth (restart_unused_codes=False), two
        batches -- the EMA statistics, the refreshed weights and the codes each depth finds in them;
      * single VQEmbedding.forward steps with the dead-code restart, on 384 vectors (>= n_embed) and on 128 (< n_embed: the
        tile-with-noise branch), torch.rand_like / torch.randperm replaced by a seeded numpy stream so that any device can
        replay the draws.  (Restarted codewords are copies of input vectors: searching them again with the same inputs is
        ambiguous at fp32 distance noise, in the reference too, so the restart is pinned one step at a time.)"""
    from rqvae.models.rqvae.quantizations import RQBottleneck, VQEmbedding
    K, Dm, seed, decay = 300, 64, 71, 0.9
    rng = np.random.default_rng(seed)
    cb = rng.standard_normal((K, Dm), dtype=np.float32)
    cs0 = rng.uniform(0.0, 3.0, K).astype(np.float32)
    xs = [rng.standard_normal((6, 8, 8, Dm), dtype=np.float32), rng.standard_normal((2, 8, 8, Dm), dtype=np.float32)]
    x_many = rng.standard_normal((384, Dm), dtype=np.float32)
    x_few = rng.standard_normal((128, Dm), dtype=np.float32)
    real = torch.randperm, torch.rand_like

    def fake(prng):
        torch.randperm = lambda n, device=None: torch.from_numpy(prng.permutation(n))
        torch.rand_like = lambda t: torch.from_numpy(prng.random(tuple(t.shape), dtype=np.float32))

    def load(vq):
        vq.weight.data.copy_(torch.from_numpy(np.concatenate([cb, np.zeros((1, Dm), np.float32)])))
        vq.embed_ema.copy_(torch.from_numpy(cb * cs0[:, None]))
        vq.cluster_size_ema.copy_(torch.from_numpy(cs0))
    out = dict(seed=seed, K=K, D=Dm, decay=np.float32(decay))
    rq = RQBottleneck([8, 8, Dm], [8, 8, 4], K, decay=decay, shared_codebook=True, restart_unused_codes=False).train()
    vq = rq.codebooks[0]
    load(vq)
    for b, x in enumerate(xs):
        quant_list, codes = rq.quantize(torch.from_numpy(x))
        out.update({f'codes{b}': codes.numpy().astype(np.int32), f'quant_last{b}': quant_list[-1].numpy(),
                    f'weight{b}': vq.weight[:-1].numpy().copy(), f'cs{b}': vq.cluster_size_ema.numpy().copy(),
                    f'ee{b}': vq.embed_ema.numpy().copy()})
    try:
        for tag, xv, sd in (('many', x_many, seed + 1), ('few', x_few, seed + 2)):
            one = VQEmbedding(K, Dm, decay=decay, restart_unused_codes=True).train()
            load(one)
            fake(np.random.default_rng(sd))
            emb, code = one(torch.from_numpy(xv))
            out.update({f'{tag}_codes': code.numpy().astype(np.int32), f'{tag}_embeds': emb.numpy(), f'{tag}_weight': one.weight[:-1].numpy().copy(),
                        f'{tag}_cs': one.cluster_size_ema.numpy().copy(), f'{tag}_ee': one.embed_ema.numpy().copy()})
    finally:
        torch.randperm, torch.rand_like = real
    # the oracle replays all of it
    w, cs, ee = cb, cs0, cb * cs0[:, None]
    for b, x in enumerate(xs):
        ql, codes, w, cs, ee = oracle.rq_quantize_train(x, w, cs, ee, 4, decay, 1e-5, None)
        assert np.array_equal(codes, out[f'codes{b}'])
        print(f'  ema[batch {b}] oracle vs ref: quants {rel(ql[-1], out[f"quant_last{b}"]):.2e}, weight {rel(w, out[f"weight{b}"]):.2e}, '
              f'cluster_size_ema {rel(cs, out[f"cs{b}"]):.2e}, embed_ema {rel(ee, out[f"ee{b}"]):.2e}')
    for tag, xv, sd in (('many', x_many, seed + 1), ('few', x_few, seed + 2)):
        rv = oracle.ema_restart_candidates(xv, K, np.random.default_rng(sd))
        quant, code, w1, cs1, ee1 = oracle.vq_ema_step(cb, cs0, cb * cs0[:, None], xv, decay, 1e-5, rv)
        assert np.array_equal(code, out[f'{tag}_codes']) and np.array_equal(quant, out[f'{tag}_embeds'])
        print(f'  ema[restart, {len(xv)} vectors] oracle vs ref: weight {rel(w1, out[f"{tag}_weight"]):.2e}, embed_ema '
              f'{rel(ee1, out[f"{tag}_ee"]):.2e}; restarted codes {(out[f"{tag}_cs"] == 1).sum()}')
    save('rq_ema.npz', **out)


def gen_param_counts():
    counts = {}
    for name in C.PARAM_COUNTS_M:
        cfg = getattr(C, name)
        with torch.device('meta'):
            m = RQTransformer(to_cfg(cfg))
        n = sum(p.numel() for p in m.parameters())
        mine = sum(int(np.prod(s)) for s in oracle.rqt_param_shapes(cfg).values())
        counts[name] = [n, mine]
        print(f'  {name}: reference {n / 1e6:.1f} M, shape table {mine / 1e6:.1f} M')
        assert n == mine
    with open(os.path.join(HERE, 'param_counts.json'), 'w') as f:
        json.dump(counts, f, indent=1)


if __name__ == '__main__':
    which = sys.argv[1:] or ['rq', 'sampler', 'vae', 'vae_batch', 'rqt', 'rqt_sample', 'rqt_big', 'rqt_var', 'ema', 'counts']
    for w in which:
        print(f'[{w}]')
        if w.startswith('rqt_big:'):                      # e.g. rqt_big:txt32,txt64
            gen_rqt_big(w.split(':', 1)[1].split(','))
            continue
        {'rq': gen_rq, 'sampler': gen_sampler, 'vae': gen_vae, 'vae_batch': gen_vae_batch, 'rqt': gen_rqt, 'rqt_sample': gen_rqt_sample_stats, 'rqt_big': gen_rqt_big, 'rqt_var': gen_rqt_variants,
         'ema': gen_ema, 'counts': gen_param_counts}[w]()
