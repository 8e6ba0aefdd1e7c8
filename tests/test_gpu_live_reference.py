"""Parity against the reference RUN LIVE (oracle/_ref: the reference's own modules as bytecode, oracle/build_ref.py; CPU, a process
of its own) on seeds that no committed fixture uses -- so that the golden tests do not hinge on the seeds they were generated
with.  Same tolerances as tests/test_gpu_parity.py.  Skipped where oracle/_ref was not built.  Run with -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle import configs as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isdir(os.path.join(ROOT, 'oracle', '_ref', 'rqvae')), reason='oracle/_ref not built')]
DEV = 'cuda:0'


def ref(case, seed, tmp_path):
    out = str(tmp_path / f'{case}_{seed}.npz')
    env = dict(os.environ)
    env['HIP_VISIBLE_DEVICES'] = ''
    env['OMP_NUM_THREADS'] = '16'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'ref_run.py'), '--case', case, '--seed', str(seed), '--out', out],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def G(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize('seed', [101, 202])
def test_live_vae_tiny(seed, tmp_path):
    from rqvae.models.rqvae import RQVAE
    g = ref('vae_tiny', seed, tmp_path)
    hps, dd = C.VAE_TINY
    vae = RQVAE(**hps, ddconfig=dd, checkpointing=False)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_params(oracle.rqvae_param_shapes(hps, dd), seed).items()})
    vae = vae.to(DEV).eval()
    err = np.abs(vae.decode_code(G(g['codes'], torch.long)).cpu().numpy() - g['decode_code'])
    print(f'live reference, vae tiny seed {seed}: decode max err {err.max():.4f} mean {err.mean():.5f}')
    assert err.max() < 0.06 and err.mean() < 0.01
    err = np.abs(vae.encode(G(g['x'])).cpu().numpy() - g['z_e'])
    assert err.max() < 0.05 and err.mean() < 0.008
    assert (vae.get_codes(G(g['x'])).cpu().numpy() == g['enc_codes']).mean() > 0.8


@pytest.mark.parametrize('seed', [303, 404])
def test_live_rqt_tiny(seed, tmp_path):
    from rqvae.models.rqtransformer import RQTransformer
    from rqvae.models.rqvae import RQVAE
    g = ref('rqt_tiny', seed, tmp_path)
    hps, dd = C.VAE_TINY
    vae = RQVAE(**hps, ddconfig=dd, checkpointing=False)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_params(oracle.rqvae_param_shapes(hps, dd), seed + 1).items()})
    vae = vae.to(DEV).eval()
    cfg = C.RQT_TINY
    ar = RQTransformer(cfg)
    ar.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_params(oracle.rqt_param_shapes(cfg), seed, cfg).items()})
    ar = ar.to(DEV).eval()
    logits = ar(G(g['codes'], torch.long), vae, cond=G(g['cond'], torch.long)).cpu().numpy()
    err = np.abs(logits - g['logits'])
    print(f'live reference, rqt tiny seed {seed}: logits max err {err.max():.4f} mean {err.mean():.5f}')
    assert err.max() < 0.06 and err.mean() < 0.01


def test_live_rq_quantize(tmp_path):
    from rqvae import _native
    g = ref('rq', 505, tmp_path)
    cb, x = G(g['cb']), G(g['x'].reshape(-1, g['x'].shape[-1]))
    codes, quants = _native.rq_quantize(x, [cb] * 4)
    clear = np.minimum.accumulate(g['gaps'] > 1e-3, axis=-1)
    assert clear.mean() > 0.99
    assert np.array_equal(codes.cpu().numpy().reshape(g['codes'].shape)[clear], g['codes'][clear])
    if clear.all():
        np.testing.assert_array_equal(quants[-1].cpu().numpy().reshape(g['quant_last'].shape), g['quant_last'])
