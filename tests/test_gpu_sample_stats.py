"""End-to-end sampling parity, statistically.  `RQTransformer.sample` cannot be compared code for code with the
reference (bf16 logits, another generator), and "every draw lies inside the filtered support" would also pass with a KV cache that
attends the wrong position.  Two tests close that:

1. Against the REFERENCE's own `sample()` (transformers.py:294-369 + utils.py:82-123, run in the build container by
   tests/golden/make_golden.py on the tiny 4 x 4 x 4 model, 20 000 images per case; fixture `rqt_tiny_sample_stats.npz`): two-sample
   chi^2 between the reference's and the HIP engine's code marginals at the first three sampled positions x 4 depths, and on two
   coarse pairwise tables (position 0 x position 1 at depth 0 -- the spatial context through the body stack and its KV cache --
   and depth 0 x depth 1 at position 0 -- the depth context through the head stack); with class conditioning, with `cond=None`,
   and from `start_loc=(1, 0)` behind a given first row.  Full softmax (a top-k / top-p boundary moves with bf16 rounding).
2. Self-consistency with the filters on (temperature 0.9, top-k 50, top-p 0.9; 4 096 images = 262 144 draws): under the engine's
   own teacher-forced logits (equal to the logits its sampler saw bit for bit, and to the reference's within 0.011 --
   test_gpu_parity.py) every draw must come from the filtered distribution q of ITS context:  sum [log q(x) + H(q)] is a zero-mean
   martingale whose variance is sum Var_q(log q); |z| < 5.  A draw from any other conditional (a stale cache row, a shifted
   position, a depth off by one, a different temperature) moves z by hundreds."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle import configs as C

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from sample_stats import SAMPLE_STATS_N, sample_stats_inputs, sample_stats_counts  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
Z = 5.0          # |z| bound of every statistic (false alarm ~ 6e-7 each)


def _models(vae_seed, rqt_seed):
    from rqvae.models.rqvae import RQVAE
    from rqvae.models.rqtransformer import RQTransformer
    hps, dd = C.VAE_TINY
    vae = RQVAE(**hps, ddconfig=dd, checkpointing=False)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_params(oracle.rqvae_param_shapes(hps, dd), vae_seed).items()})
    ar = RQTransformer(C.RQT_TINY)
    ar.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_params(oracle.rqt_param_shapes(C.RQT_TINY), rqt_seed).items()})
    return vae.to(DEV).eval(), ar.to(DEV).eval()


def chi2_two_sample(a, b):
    """two-sample chi^2 of two count vectors with equal totals; cells with fewer than 10 counts together are pooled"""
    a, b = a.astype(np.float64).ravel(), b.astype(np.float64).ravel()
    keep = (a + b) >= 10
    aa, bb = np.append(a[keep], a[~keep].sum()), np.append(b[keep], b[~keep].sum())
    nz = (aa + bb) > 0
    stat = float((((aa - bb) ** 2)[nz] / (aa + bb)[nz]).sum())
    dof = int(nz.sum()) - 1
    return stat, dof, (stat - dof) / np.sqrt(2.0 * dof)


def _sample_all(ar, vae, part, cond, start, seed, chunk=5000, **kw):
    out = []
    for i in range(0, part.shape[0], chunk):
        torch.cuda.manual_seed_all(seed + i)
        c = None if cond is None else torch.from_numpy(cond[i:i + chunk]).to(DEV)
        out.append(ar.sample(torch.from_numpy(part[i:i + chunk]).to(DEV), vae, cond=c, start_loc=start, **kw).cpu().numpy())
    return np.concatenate(out)


@pytest.mark.parametrize('case', ['cond', 'nocond', 'start'])
def test_sample_statistics_vs_reference_sample(golden, case):
    g = golden('rqt_tiny_sample_stats.npz')
    assert int(g['n']) == SAMPLE_STATS_N
    vae, ar = _models(int(g['vae_seed']), int(g['seed']))
    cfg = C.RQT_TINY
    part, cond, start = sample_stats_inputs(cfg, case)
    xs = _sample_all(ar, vae, part, cond, start, seed=4242)
    if case == 'start':
        assert np.array_equal(xs[:, 0], part[:, 0])                      # the given first row is kept (transformers.py:346-349)
    marg, p_sp, p_dp = sample_stats_counts(xs, cfg, start)
    worst = 0.0
    for i in range(3):
        for d in range(cfg['block_size'][2]):
            stat, dof, z = chi2_two_sample(g[f'marg_{case}'][i, d], marg[i, d])
            worst = max(worst, z)
            assert z < Z, f'{case}: marginal of sampled position {i}, depth {d}: chi^2 {stat:.1f} on {dof} dof (z = {z:.1f})'
    zs = []
    for name, ref, got in (('position 0 x position 1 (depth 0)', g[f'pair_spatial_{case}'], p_sp),
                           ('depth 0 x depth 1 (position 0)', g[f'pair_depth_{case}'], p_dp)):
        stat, dof, z = chi2_two_sample(ref, got)
        zs.append(z)
        assert z < Z, f'{case}: pairwise table {name}: chi^2 {stat:.1f} on {dof} dof (z = {z:.1f})'
    # the test has power: the same statistic between DIFFERENT depths of the engine's own samples is far outside the bound
    _, _, z_power = chi2_two_sample(marg[0, 0], marg[0, 1])
    print(f'sample statistics [{case}]: 12 marginals worst z {worst:+.2f}, pairwise tables z {zs[0]:+.2f} / {zs[1]:+.2f} '
          f'(bound {Z}); depth 0 vs depth 1 of the same samples z {z_power:.0f}')
    assert z_power > Z


def test_sample_draws_follow_the_filtered_conditionals(golden):
    g = golden('rqt_tiny_sample_stats.npz')
    vae, ar = _models(int(g['vae_seed']), int(g['seed']))
    cfg = C.RQT_TINY
    n, T, K, P = 4096, 0.9, 50, 0.9
    part, cond, start = sample_stats_inputs(cfg, 'cond', n)
    xs = _sample_all(ar, vae, part, cond, start, seed=99, chunk=n, temperature=T, top_k=K, top_p=P)
    logits = ar(torch.from_numpy(xs).to(DEV), vae, cond=torch.from_numpy(cond).to(DEV)).cpu().numpy()       # (n, H, W, D, V) teacher-forced
    rows = logits.reshape(-1, logits.shape[-1]).astype(np.float32)
    draws = xs.reshape(-1)

    def z_of(q, draws):
        qx = q[np.arange(q.shape[0]), draws]
        inside = qx > 0
        with np.errstate(divide='ignore', invalid='ignore'):
            lq = np.where(q > 0, np.log(q), 0.0)
        ent = -(q * lq).sum(-1)
        var = (q * lq * lq).sum(-1) - ent ** 2
        s = (np.log(qx[inside]) + ent[inside]).sum()
        return float(s / np.sqrt(var[inside].sum())), int((~inside).sum())

    q = np.concatenate([oracle.sampler.filtered_probs(rows[i:i + 16384], temperature=T, top_k=K, top_p=P) for i in range(0, rows.shape[0], 16384)])
    z, outside = z_of(q, draws)
    # power: the same draws against the conditionals of the NEXT depth's context (what an off-by-one in the depth / cache index would
    # sample from), and against another temperature
    q_shift = q.reshape(n, -1, q.shape[-1])
    z_shift, out_shift = z_of(np.roll(q_shift, 1, axis=1).reshape(q.shape), draws)
    q_t = np.concatenate([oracle.sampler.filtered_probs(rows[i:i + 16384], temperature=1.0, top_k=K, top_p=P) for i in range(0, rows.shape[0], 16384)])
    z_t, _ = z_of(q_t, draws)
    print(f'sampling self-consistency (T {T}, top-k {K}, top-p {P}; {draws.size} draws): z = {z:+.2f} (bound {Z}), {outside} draws outside '
          f'the filtered support; against the neighbouring context z = {z_shift:+.0f} ({out_shift} outside), against temperature 1.0 z = {z_t:+.0f}')
    assert outside <= draws.size // 20000, f'{outside} draws outside the filtered support of their own context'
    assert abs(z) < Z
    assert abs(z_shift) > 10 * Z or out_shift > draws.size // 10
    assert abs(z_t) > 2 * Z
