#!/usr/bin/env python
"""What shader clock does the GPU sustain under each of the step's dominant kernels?  (round 5)  The dense MFMA peak the roofline is priced
against (2.5 PFLOP/s bf16) is 256 CUs x 4 SIMDs x 1024 FLOP per clock at 2.4 GHz; a kernel that keeps the matrix pipes AND the LDS / L2 /
HBM paths busy may run below that clock under the board's power limit, which caps what any schedule of the same instructions can reach.
Runs each kernel back to back for ~3 s while a background thread samples `rocm-smi` (sclk, power), and reports the medians of the last 2 s.
    python scripts/clock_under_load.py"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import _native

dev = 'cuda'
samples, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        t = time.time()
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:          # noqa: BLE001
            out = str(e)
        m = re.search(r'sclk clock level: *\S+ *\((\d+) ?Mhz\)', out, re.I)
        p = re.search(r'Power \(W\): *([\d.]+)', out)
        samples.append((t, int(m.group(1)) if m else None, float(p.group(1)) if p else None, out if not m else ''))
        time.sleep(0.1)


def med(v):
    v = sorted(x for x in v if x is not None)
    return v[len(v) // 2] if v else None


def load(name, fn, flop, secs=3.0):
    fn(); torch.cuda.synchronize()
    samples.clear()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.time() - t0 < secs:
        if time.time() - t0 > secs - 1.0 and n >= 0:
            e0.record()
            for _ in range(50):
                fn()
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            n = -1
        else:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
    late = [s for s in samples if s[0] - t0 > 1.0]
    sclk, pw = med([s[1] for s in late]), med([s[2] for s in late])
    peak = 2.5e3 * sclk / 2400.0 if sclk else float('nan')
    tf = flop / us / 1e6 if flop else 0.0
    print(f'{name:44s} {us:9.1f} us {tf:8.1f} TF | sclk {sclk} MHz, {pw} W ({len(late)} samples) | dense bf16 peak at that clock {peak:6.0f} TF'
          + (f' -> {tf / peak:.3f} of it' if flop else ''), flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(1.0)
# the matrix pipes alone (scripts/micro/mfma_power.hip: registers only, no LDS or memory traffic), by operand data
exe = os.path.join(ROOT, 'scripts', 'micro', 'bin', 'mfma_power')
if os.path.exists(exe):
    for mode, what in ((0, 'constant operands'), (1, 'operands change every MFMA, ~N(0,1)'), (2, 'one operand all zeros')):
        samples.clear()
        t0 = time.time()
        r = subprocess.run([exe, str(mode), '3'], capture_output=True, text=True).stdout.strip()
        late = [s for s in samples if s[0] - t0 > 1.0]
        print(f'MFMAs only, {what:38s}: {r} | sclk {med([s[1] for s in late])} MHz, {med([s[2] for s in late])} W', flush=True)
print('idle:', med([s[1] for s in samples]), 'MHz', med([s[2] for s in samples]), 'W', (samples[-1][3][:300] if samples and samples[-1][1] is None else ''))
g = torch.Generator(device=dev).manual_seed(0)
M = 10752
for nm, N, K, epi in (('gemm_p8 qkv', 4608, 1536, 0), ('gemm_p8 fc1 (GELU)', 6144, 1536, 1), ('gemm_p8 fc2-shaped (bf16 out)', 1536, 6144, 0)):
    a = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    w = (0.05 * torch.randn((N, K), device=dev, generator=g)).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev, generator=g)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    load(f'{nm} M={M} N={N} K={K}', lambda: _native.dbg_gemm(a, w, bias, epi, 256, 256, 1, out=out), 2.0 * M * N * K)
    az = torch.zeros_like(a)
    load(f'   the same on all-zero activations', lambda: _native.dbg_gemm(az, w, bias, epi, 256, 256, 1, out=out), 2.0 * M * N * K)
B, H, C = 32, 256, 128
x = torch.randn((B, H, H, C), device=dev, generator=g).to(torch.bfloat16)
w = (0.05 * torch.randn((C, 3, 3, C), device=dev, generator=g)).to(torch.bfloat16)
bias = torch.randn((C,), device=dev, generator=g)
resid = torch.randn((B, H, H, C), device=dev, generator=g).to(torch.bfloat16)
gn = torch.stack([1 + 0.2 * torch.randn((B, C), device=dev, generator=g), 0.3 * torch.randn((B, C), device=dev, generator=g)], -1).contiguous()
out = torch.empty_like(resid)
fl = 2.0 * B * H * H * C * 9 * C
load('halo conv 128->128 @256^2 x32, plain', lambda: _native.dbg_conv_halo(x, w, bias, out=out), fl)
load('halo conv 128->128 @256^2 x32, GN+SiLU+resid', lambda: _native.dbg_conv_halo(x, w, bias, gn=gn, resid=resid, out=out), fl)
stop.set()
