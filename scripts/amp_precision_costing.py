#!/usr/bin/env python
"""CPU / build container only (imports the reference from /root/reference): what the reference's own `amp=True` costs in logits, next to this
engine's bf16.  The reference's fp16 path is `torch.cuda.amp.autocast` (transformers.py:21,114,206; main_sampling_fid.py:216); there is no GPU
here, so the module's `autocast` name is pointed at `torch.autocast('cpu', dtype=...)` -- the CPU autocast policy (matmuls in the low
precision, LayerNorm / softmax in fp32), a stand-in for the CUDA one.  Prints max / mean |logits - fp32 logits| for fp16 and bf16 autocast on
the same seeded weights and codes as the fixtures `rqt_wide` (E 1536 / 24 heads, 2 + 1 layers) and `rqt_tiny`; the engine's own error on those
two shapes is in profiles/r05_pytest_gpu_final.log (VERDICT r04 "missing" item 5)."""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_m = types.ModuleType('omegaconf')
_m.OmegaConf = type('OmegaConf', (), {})
_m.MISSING = '???'
_m.DictConfig = dict
sys.modules['omegaconf'] = _m
sys.path.insert(0, '/root/reference')
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from rqvae.models.rqtransformer import RQTransformer  # noqa: E402  (reference)
import rqvae.models.rqtransformer.transformers as T  # noqa: E402  (reference)
import oracle  # noqa: E402
from oracle import configs as C  # noqa: E402

torch.set_grad_enabled(False)


class Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def copy(self):
        return to_cfg(json.loads(json.dumps(self)))


def to_cfg(d):
    return Cfg({k: to_cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


class Aux:
    def __init__(self, cb):
        self.cb = torch.from_numpy(cb)

    def get_code_emb_with_depth(self, xs):
        return F.embedding(xs, self.cb), None


cuda_autocast = T.autocast
for tag, cfg, seed, dim in (('rqt_tiny (E 128, 2 + 2 layers, V 500)', C.RQT_TINY, 41, 64), ('rqt_wide (E 1536, 2 + 1 layers, V 16384)', C.RQT_WIDE, 45, 256)):
    m = RQTransformer(to_cfg(cfg)).eval()
    params = oracle.make_params(oracle.rqt_param_shapes(cfg), seed, cfg)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    rng = np.random.default_rng(1)
    V, (H, W, D) = cfg['vocab_size'], cfg['block_size']
    aux = Aux(rng.standard_normal((V, dim), dtype=np.float32))
    codes = torch.from_numpy(rng.integers(0, V, (2, H, W, D)))
    cond = torch.from_numpy(rng.integers(0, cfg['vocab_size_cond'], (2, 1)))
    T.autocast = cuda_autocast
    ref = m(codes, aux, cond=cond).numpy()
    print(f'{tag}: |logits| max {np.abs(ref).max():.2f} std {ref.std():.3f}')
    for name, dt in (('fp16', torch.float16), ('bf16', torch.bfloat16)):
        T.autocast = lambda enabled=True, dt=dt: torch.autocast('cpu', dtype=dt, enabled=enabled)
        out = m(codes, aux, cond=cond, amp=True).float().numpy()
        err = np.abs(out - ref)
        print(f'   reference, amp=True as {name} autocast (CPU policy): max {err.max():.4f} mean {err.mean():.5f}; top-1 agreement {float((out.argmax(-1) == ref.argmax(-1)).mean()):.3f}')
