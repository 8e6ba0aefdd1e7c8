#!/usr/bin/env python
"""The drivers' one-image-per-call decode loop: graph replay vs eager launches (RQAMD_VAE_GRAPH=0), host and device time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import presets  # noqa: E402

dev = torch.device('cuda', 0)
torch.set_grad_enabled(False)
vae, _ = None, None
from rqvae.models import create_model  # noqa: E402
from rqvae.utils.config import Config, augment_arch_defaults  # noqa: E402
import copy  # noqa: E402
torch.manual_seed(0)
with dev:
    vae, _ = create_model(augment_arch_defaults(Config(copy.deepcopy(presets.RQVAE['imagenet']))))
vae = vae.eval()
n = int(os.environ.get('RQ_N', 64))
codes = torch.randint(0, 16384, (n, 8, 8, 4), device=dev)
for i in range(4):
    vae.decode_code(codes[i:i + 1])
torch.cuda.synchronize()
t0 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
pix = torch.cat([vae.decode_code(codes[i:i + 1]) for i in range(n)], dim=0)
e1.record()
t1 = time.perf_counter()
e1.synchronize()
t2 = time.perf_counter()
print(f'graph={os.environ.get("RQAMD_VAE_GRAPH", "1")}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/img, device {e0.elapsed_time(e1) / n:.3f} ms/img, wall {1e3 * (t2 - t0) / n:.3f} ms/img')
x = torch.randn((8, 3, 256, 256), device=dev).clamp(-1, 1)
for nm, fn in (('encode', lambda i: vae.encode(x[i:i + 1])), ('quantize', lambda i: vae.quantizer(z[i:i + 1])), ('decode', lambda i: vae.decode(zq[i:i + 1]))):
    z = vae.encode(x)
    zq = vae.quantizer(z)[0]
    fn(0)
    torch.cuda.synchronize()
    e0.record()
    for i in range(8):
        fn(i)
    e1.record()
    e1.synchronize()
    print(f'  per-image {nm}: {e0.elapsed_time(e1) / 8:.3f} ms')
