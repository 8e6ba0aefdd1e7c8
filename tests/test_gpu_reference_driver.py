"""The reference's UNCHANGED throughput driver on the MI355X: `measure_throughput/__main__.py`, byte-compiled
from /root/reference by oracle/build_ref.py (oracle/_ref travels to the GPU box; /root/reference does not), launched through
rq-vae-transformer_amd/rqamd_run.py -- which only arranges sys.path: this repo's `rqvae` first, oracle/_ref second -- builds the
1.4B RQ-Transformer + RQ-VAE through ITS create_model, moves them to cuda and runs ITS timed loops (`model_ar.sample(...)`, then
`torch.cat([model_aux.decode_code(chunk) for chunk in codes.chunk(batch_size)])`, :293-301).  Nothing of the driver is patched;
`omegaconf` / `easydict`, which this image lacks, come from tests/stubs (test infrastructure).  Checked: it completes, prints its
summary line, and its per-image decode runs at the batched rate (the read-ahead behind RQVAE.decode_code).  Run with -m gpu."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, 'oracle', '_ref')
pytestmark = pytest.mark.gpu


def test_unchanged_measure_throughput_runs_its_loop_on_the_gpu():
    if not os.path.exists(os.path.join(REF, 'measure_throughput', '__main__.pyc')):
        pytest.skip('oracle/_ref not built (python oracle/build_ref.py, in the build container)')
    env = dict(os.environ)
    env['PYTHONPATH'] = os.path.join(ROOT, 'tests', 'stubs')
    env['RQVAE_REFERENCE_ROOT'] = REF
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'rq-vae-transformer_amd', 'rqamd_run.py'), '-m', 'measure_throughput',
                        'model=huge', 'f=32', 'd=4', 'c=16384', 'batch_size=200', 'n_loop=2', 'warmup=1'],
                       capture_output=True, text=True, env=env, cwd=REF, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    m = re.search(r'\| ([0-9.]+) ms/sample \(ar: ([0-9.]+), decode: ([0-9.]+)\)\s*\n=+', r.stdout)
    assert m, r.stdout[-2000:]
    total, ar_ms, dec_ms = (float(x) for x in m.groups())
    print(f'unchanged measure_throughput (1.4B, 8x8x4, batch 200, fp32-API / bf16 engine): {total:.3f} ms/sample (ar {ar_ms:.3f}, decode {dec_ms:.3f}) '
          f'= {1e3 / total:.0f} images/s')
    assert 'rqgan size: 10' in r.stdout and 'rqtransformer size: 13' in r.stdout          # 104.4 M and 1.39 B parameters, as the script counts them
    assert dec_ms < 0.6, dec_ms            # one decode_code call per image, served at the batched rate (2.0 ms per cold call)
    assert total < 4.0, total              # batch 200: ~2.1 ms/sample
