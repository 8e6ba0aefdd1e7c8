#!/usr/bin/env python
"""Phase timeline of the persistent stack kernel (needs librqamd_trace.so from scripts/stack_trace.sh): RQTransformer.sample at
RQ_B rows, then the stamps of the last body launch (position 63): per step, phase time and barrier wait of two workgroups."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
os.environ['RQAMD_STACK'] = '1'
os.environ.setdefault('RQAMD_STACK_ROWS', '256')
import torch
from rqvae import _native
_native.LIB_PATH = os.path.join(ROOT, 'rq-vae-transformer_amd', 'librqamd_trace.so')
_native.lib()
from rqvae import presets
dev = torch.device('cuda:0')
vae, ar, cfg = presets.build(os.environ.get('RQ_PRESET', 'small'), device=dev, seed=0)
B = int(os.environ.get('RQ_B', 64))
ps = torch.zeros((B, 8, 8, 4), dtype=torch.long, device=dev)
cd = torch.zeros((B, 1), dtype=torch.long, device=dev)
for _ in range(2):
    ar.sample(ps, model_aux=vae, cond=cd, top_k=1024, top_p=0.95)
torch.cuda.synchronize()
fn = C.CDLL(_native.LIB_PATH).rqamd_dbg_stack_trace
fn.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (2 * 64 * 3))()
assert fn(buf) == 0
names = ['ln1', 'qkv', 'attn', 'proj', 'ln2', 'fc1', 'fc2']
print(f'B={B}: step: phase us | barrier us   (workgroup 0 / traced workgroup); 10 ns ticks')
tot = [0.0] * 7; totb = [0.0] * 7
for s in range(7, 35):
    row = f'block {s // 7} {names[s % 7]:5s}'
    for w in range(2):
        t = [buf[(w * 64 + s) * 3 + k] for k in range(3)]
        row += f' | {(t[1] - t[0]) / 100:6.2f} {(t[2] - t[1]) / 100:6.2f}'
        if w == 0:
            tot[s % 7] += (t[1] - t[0]) / 100 / 4; totb[s % 7] += (t[2] - t[1]) / 100 / 4
    print(row)
print('mean over 4 blocks (workgroup 0): ' + ' '.join(f'{names[i]} {tot[i]:.2f}+{totb[i]:.2f}' for i in range(7)) + f' = {sum(tot) + sum(totb):.1f} us per block')
