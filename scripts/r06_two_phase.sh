#!/bin/bash
# round 6: two-phase RQ-VAE calls: parity test, then decode / get_codes throughput with and without (RQAMD_VAE_TWO_PHASE=0), same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_phase or small_chunk or batch_invariance" 2>&1 | tail -3
cat > /tmp/tp.py <<'PY'
import os, sys, torch
sys.path.insert(0, 'rq-vae-transformer_amd')
from rqvae import presets
dev = torch.device('cuda:0')
vae, ar, cfg = presets.build('huge', device=dev, seed=0)
del ar
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for B in (256, 512, 1024):
    codes = torch.randint(0, 16384, (B, 8, 8, 4), device=dev)
    x = torch.randn((min(B, 512), 3, 256, 256), device=dev).clamp(-1, 1)
    d = t(lambda: vae.decode_code(codes)); e = t(lambda: vae.get_codes(x))
    print(f"TWO_PHASE={os.environ.get('RQAMD_VAE_TWO_PHASE', '1')} B={B}: decode {d / B:.4f} ms/img | get_codes({x.shape[0]}) {e:.2f} ms = {x.shape[0] / e * 1e3:.0f} img/s", flush=True)
PY
for tp in 1 0 1 0; do RQAMD_VAE_TWO_PHASE=$tp python /tmp/tp.py 2>&1 | grep -v amdgpu; done
