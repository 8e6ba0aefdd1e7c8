#!/usr/bin/env python
"""Upsample.conv of the decoder (nearest 2x + 3x3 conv) at the released shapes: the 9-tap folded-upsample form (persistent halo kernel) against
the sub-pixel form (four 2 x 2 convs over the source image, 4 taps per output pixel), time per launch and difference of the results."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native  # noqa: E402

dev = 'cuda'
B = int(os.environ.get('RQ_B', 128))
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for Hs, C in ((32, 256), (64, 256), (128, 128)):
    xs = torch.randn((B, Hs, Hs, C), device=dev, generator=g).to(torch.bfloat16)
    w = (0.03 * torch.randn((C, 3, 3, C), device=dev, generator=g)).to(torch.bfloat16)
    bias = torch.randn((C,), device=dev, generator=g)
    H = 2 * Hs
    stats = torch.zeros((B, (H // 8) * (H // 32), 32, 2), device=dev)
    a = _native.dbg_conv_halo(xs, w, bias, ups=True, stats=stats)
    wsub = _native.dbg_ups_subpixel_weights(w)
    o2 = torch.empty_like(a)

    def sub():
        # (the engine prepares the pre-summed weights once per checkpoint: time the conv alone)
        _native.check(_native.lib().rqamd_dbg_conv_halo_bf16(_native.ptr(xs, torch.bfloat16), _native.ptr(wsub, torch.bfloat16), _native.ptr(bias, torch.float32), None, None,
                                                             B, H, H, C, C, 1 | 64, _native.ptr(o2), _native.ptr(stats), _native.stream_of(xs)))
    sub()
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(xs[:4].float().permute(0, 3, 1, 2), scale_factor=2.0, mode='nearest'),
                                     w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    e_a = ((a[:4].float() - ref).abs().max() / ref.abs().max()).item()
    e_b = ((o2[:4].float() - ref).abs().max() / ref.abs().max()).item()
    res = []
    for rep in range(2):
        t_a = timed(lambda: _native.dbg_conv_halo(xs, w, bias, ups=True, stats=stats, out=a))
        t_b = timed(sub)
        res.append(f'folded {t_a:7.1f} us  sub-pixel {t_b:7.1f} us')
    fl = 2.0 * B * H * H * C * C * 9
    print(f'B={B} {Hs}^2 -> {H}^2, {C} ch: ' + ' | '.join(res) + f'  ({fl / t_a / 1e6:.0f} -> {fl / t_b / 1e6:.0f} TFLOP/s of the 9-tap FLOPs); max err vs torch fp32: '
          f'folded {e_a:.4f}, sub-pixel {e_b:.4f} of max', flush=True)
