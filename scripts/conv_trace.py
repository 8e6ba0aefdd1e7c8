#!/usr/bin/env python
"""Per-tap barrier timeline of one workgroup of the 8-row halo conv (needs librqamd_trace.so from scripts/conv_trace.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch
from rqvae import _native
_native.LIB_PATH = os.path.join(ROOT, 'rq-vae-transformer_amd', os.environ.get('RQ_TRACE_LIB', 'librqamd_trace.so'))
lib = _native.lib()
dev = 'cuda'
B, H, Cin, Cout = int(os.environ.get('RQ_B', 8)), int(os.environ.get('RQ_H', 256)), 128, 128
x = torch.randn((B, H, H, Cin), device=dev).to(torch.bfloat16)
w = (torch.randn((Cout, 3, 3, Cin), device=dev) * 0.05).to(torch.bfloat16)
bias = torch.randn((Cout,), device=dev)
resid = torch.randn((B, H, H, Cout), device=dev).to(torch.bfloat16)
gn = torch.stack([1 + 0.2 * torch.randn((B, Cin), device=dev), 0.3 * torch.randn((B, Cin), device=dev)], -1).contiguous()
out = torch.empty_like(resid)
st = torch.zeros((B, (H // 8) * (H // 32), 32, 2), device=dev)
buf = (C.c_ulonglong * 512)()
fn = C.CDLL(_native.LIB_PATH).rqamd_dbg_conv_trace
fn.argtypes = [C.POINTER(C.c_ulonglong)]
PERS = os.environ.get('RQ_PERSIST', '0') == '1'
for name, kw in (('plain', {}), ('GN+resid+stats', dict(gn=gn, resid=resid, stats=st))):
    for rep in range(3):
        _native.dbg_conv_halo(x, w, bias, out=out, persistent=PERS, **kw)
    torch.cuda.synchronize()
    assert fn(buf) == 0
    t = [[buf[wv * 64 + i] for i in range(64)] for wv in range(8)]
    print(f'== {name} persistent={PERS}: cycles relative to wave 0 prologue start; per tap: compute (release->arrive) / barrier wait (arrive->release)')
    t00 = t[0][0]
    for wv in (0, 1, 3, 4, 7):
        r = t[wv]
        line = f'wave {wv}: prologue {r[1] - r[0]:6d} |'
        comp, wait = [], []
        for k in range(18):
            arrive, release = r[2 + 2 * k], r[3 + 2 * k]
            prev_release = r[1] if k == 0 else r[3 + 2 * (k - 1)]
            comp.append(arrive - prev_release); wait.append(release - arrive)
        line += ' taps: ' + ' '.join(f'{c}/{w_}' for c, w_ in zip(comp, wait))
        line += f' | epilogue: to pack barrier {r[61] - r[37]:6d}, to end {r[63] - r[37]:6d} | tile total {r[63] - r[0]:6d}'
        print(line)
    print(f'   clock: {(t[0][63] - t[0][0]) / max(1, (t[0][59] - t[0][58])) * 100:.0f} MHz shader ticks (s_memtime) per 100 MHz s_memrealtime')
    print(f'   mean compute {sum(sum(t[wv][2 + 2 * k] - (t[wv][1] if k == 0 else t[wv][1 + 2 * k]) for k in range(18)) for wv in range(8)) / 144:.0f}'
          f' mean wait {sum(sum(t[wv][3 + 2 * k] - t[wv][2 + 2 * k] for k in range(18)) for wv in range(8)) / 144:.0f}')
