#!/usr/bin/env python
"""GPU microbenchmark of the residual quantiser (rqamd_rq_quantize) and of RQVAE.get_codes
(encode + quantise): codes/sec, TFLOP/s against the fp32 MFMA peak, algorithmic GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

dev = 'cuda'


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


print('residual quantiser, K=16384 D=256 depth=4 (ImageNet RQ-VAE codebook), N = 64 vectors per image')
gen = torch.Generator(device=dev).manual_seed(0)
cb = torch.randn((16384, 256), device=dev, generator=gen)
for B in (1, 4, 16, 64, 256, 1024):
    x = torch.randn((B * 64, 256), device=dev, generator=gen)
    ms = timeit(lambda: _native.rq_quantize(x, [cb] * 4, want_quants=True), 3 if B >= 256 else 10)
    flops = 2.0 * B * 64 * 16384 * 256 * 4
    bytes_ = B * 133120 + 16384 * 256 * 4
    print(f'  B={B:5d}: {ms:8.3f} ms  {B / ms * 1e3:9.0f} img/s  {B * 256 / ms * 1e3 / 1e6:7.2f} Mcodes/s  '
          f'{flops / ms / 1e9:6.1f} TFLOP/s fp32 ({flops / ms / 1e9 / 157.3 * 100:4.1f}% of 157.3)  '
          f'{bytes_ / ms / 1e6:7.1f} GB/s algorithmic ({bytes_ / ms / 1e6 / 8000 * 100:.2f}% of HBM)', flush=True)

print('RQVAE.get_codes (encode 256x256 + quantise), ImageNet RQ-VAE shape, random-init weights')
from rqvae import presets  # noqa: E402
from rqvae.models.rqvae import RQVAE  # noqa: E402
hps, dd = presets.RQVAE["imagenet"]["hparams"], presets.RQVAE["imagenet"]["ddconfig"]
torch.manual_seed(0)
with torch.device(dev):
    vae = RQVAE(**hps, ddconfig=dd, checkpointing=False).eval()
for B in (64, 256):
    x = torch.randn((B, 3, 256, 256), device=dev).clamp(-1, 1)
    ms = timeit(lambda: vae.get_codes(x), 2)
    ms_enc = timeit(lambda: vae.encode(x), 2)
    print(f'  B={B:4d}: get_codes {ms:8.2f} ms ({B / ms * 1e3:7.0f} img/s, {B * 256 / ms * 1e3 / 1e6:6.3f} Mcodes/s); '
          f'encode alone {ms_enc:8.2f} ms ({134.2 * B / ms_enc:6.1f} TFLOP/s conv)', flush=True)
