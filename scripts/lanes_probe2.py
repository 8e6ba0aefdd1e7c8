#!/usr/bin/env python
"""Probe (round 6): L independent decode chains ("lanes") of B/L rows each on L HIP streams, ALL ON ONE COPY OF THE WEIGHTS
(rqamd_dbg_rqt_share_params), against one chain of B rows.  Round 2's lanes_probe.py ran separate engines with separate weight copies
(2.8 GB each: no L2 / MALL sharing) on the round-2 kernels and was slower everywhere; the mid-batch tiles of round 5 leave 64-112
of the 256 CUs idle per GEMM launch, which a second chain can use.  RQ_BS=64,100,200,500  RQ_LANES=1,2,3,4  RQ_MODEL=huge."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
import torch  # noqa: E402
from rqvae import _native, presets  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda', 0)
model = os.environ.get('RQ_MODEL', 'huge')
batches = [int(b) for b in os.environ.get('RQ_BS', '64,100,200,500').split(',')]
lanes = [int(b) for b in os.environ.get('RQ_LANES', '1,2,3,4').split(',')]
LMAX = max(lanes)
vae, ar, cfg = presets.build(model, device=dev, seed=0)
part = torch.zeros((2,) + tuple(ar.block_size), device=dev, dtype=torch.long)
ar.sample(part, model_aux=vae, cond=torch.zeros((2, ar.block_size_cond), device=dev, dtype=torch.long), top_k=1024, top_p=0.95)
torch.cuda.synchronize()
eng0 = ar._eng()
lib = _native.lib()
c = ar.config
engs = [eng0]
for i in range(1, LMAX):
    e = _native.RqtEngine(embed_dim=c.embed_dim, n_head=c.body.block.n_head, n_layer_body=c.body.n_layer, n_layer_head=c.head.n_layer,
                          vocab_size=max(ar.vocab_size), input_embed_dim=c.input_embed_dim, vocab_size_cond=ar.vocab_size_cond,
                          block_size_cond=ar.block_size_cond, block_size=list(ar.block_size), gelu_v2=c.body.block.gelu == 'v2', device=dev,
                          vocab_sizes=ar.vocab_size)
    _native.check(lib.rqamd_dbg_rqt_share_params(e._h, eng0._h))
    engs.append(e)
streams = [torch.cuda.Stream(dev) for _ in range(LMAX)]
cbs = ar._checked_codebooks(vae)
D = ar.block_size[2]


def run(B, L, seed=5):
    rows = [B // L + (1 if i < B % L else 0) for i in range(L)]
    ins = [(torch.zeros((r,) + tuple(ar.block_size), dtype=torch.long, device=dev), torch.zeros((r, ar.block_size_cond), dtype=torch.long, device=dev)) for r in rows]

    def once():
        cur = torch.cuda.current_stream(dev)
        outs = []
        for i in range(L):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs.append(engs[i].sample(ins[i][0], ins[i][1], cbs, (0, 0), 1.0, [1024] * D, [0.95] * D, seed, 1000 * i, True))
        for i in range(L):
            cur.wait_stream(streams[i])
        return outs
    once()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        once()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


for B in batches:
    line = f'{model} B={B:5d}:'
    for L in lanes:
        t = run(B, L)
        line += f'  L={L}: {t:7.1f} ms ({B / t * 1e3:7.1f} img/s AR only)'
    print(line, flush=True)
