#!/usr/bin/env python
"""Run the REFERENCE itself (oracle/_ref, see oracle/build_ref.py) on CPU for a seeded case and save its outputs (TEST
INFRASTRUCTURE: tests/test_gpu_live_reference.py compares the HIP path against them, on seeds no committed fixture uses).  A process
of its own: the reference's package is also called `rqvae`.  Weights come from oracle.weights.make_params (seeded, keyed by
state_dict name) exactly as tests/golden/make_golden.py loads them.
    python oracle/ref_run.py --case vae_tiny|rqt_tiny|rq --seed S --out file.npz"""
import argparse
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--case', required=True)
    ap.add_argument('--seed', type=int, required=True)
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    stub = types.ModuleType('omegaconf')
    stub.OmegaConf = type('OmegaConf', (), {})
    stub.MISSING = '???'
    stub.DictConfig = dict
    sys.modules['omegaconf'] = stub
    sys.path.insert(0, ROOT)                       # oracle (numpy only; never imports rqvae)
    sys.path.insert(0, os.path.join(HERE, '_ref'))  # the reference's rqvae (sourceless .pyc)
    import torch
    import oracle
    from oracle import configs as C
    from rqvae.models.rqvae import RQVAE
    from rqvae.models.rqtransformer import RQTransformer
    assert os.path.join('oracle', '_ref') in sys.modules['rqvae'].__file__
    torch.set_grad_enabled(False)
    rng = np.random.default_rng(a.seed + 1000)

    class Cfg(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

        def copy(self):
            return to_cfg(json.loads(json.dumps(self)))

    def to_cfg(d):
        return Cfg({k: to_cfg(v) if isinstance(v, dict) else v for k, v in d.items()})

    def vae(cfg, seed):
        hps, dd = cfg
        m = RQVAE(**hps, ddconfig=dd, checkpointing=False).eval()
        params = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), seed)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        return m
    out = {}
    if a.case == 'vae_tiny':
        m = vae(C.VAE_TINY, a.seed)
        codes = rng.integers(0, C.VAE_TINY[0]['n_embed'], (3, 8, 8, 4))
        x = np.clip(rng.standard_normal((3, 3, 16, 16), dtype=np.float32), -1, 1)
        out = dict(codes=codes, x=x, decode_code=m.decode_code(torch.from_numpy(codes)).numpy(), z_e=m.encode(torch.from_numpy(x)).numpy(),
                   enc_codes=m.get_codes(torch.from_numpy(x)).numpy())
    elif a.case == 'rqt_tiny':
        m_aux = vae(C.VAE_TINY, a.seed + 1)
        cfg = C.RQT_TINY
        ar = RQTransformer(to_cfg(cfg)).eval()
        ar.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_params(oracle.rqt_param_shapes(cfg), a.seed, cfg).items()}, strict=True)
        codes = rng.integers(0, cfg['vocab_size'], (3, 4, 4, 4))
        cond = rng.integers(0, cfg['vocab_size_cond'], (3, 1))
        out = dict(codes=codes, cond=cond, logits=ar(torch.from_numpy(codes), m_aux, cond=torch.from_numpy(cond)).numpy())
    elif a.case == 'rq':
        from rqvae.models.rqvae.quantizations import RQBottleneck
        K, Dm = 1500, 128
        cb = rng.standard_normal((K, Dm), dtype=np.float32)
        x = rng.standard_normal((5, 8, 8, Dm), dtype=np.float32)
        rq = RQBottleneck([8, 8, Dm], [8, 8, 4], K, shared_codebook=True).eval()
        rq.codebooks[0].weight.data.copy_(torch.from_numpy(np.concatenate([cb, np.zeros((1, Dm), np.float32)])))
        ql, codes = rq.quantize(torch.from_numpy(x))
        gaps, _ = oracle.rq_quantize_margins(x, [cb] * 4)
        out = dict(cb=cb, x=x, codes=codes.numpy(), quant_last=ql[-1].numpy(), gaps=gaps.astype(np.float32))
    else:
        raise SystemExit(f'unknown case {a.case}')
    np.savez(a.out, **out)
    print('ok', a.case, a.seed)


if __name__ == '__main__':
    main()
