"""oracle/_ref -- the reference's own modules, byte-compiled by oracle/build_ref.py (test infrastructure; bench.py's cpu_baseline
leg, kind "reference"): the recipe builds when /root/reference is present, the result imports sourceless in a process of its
own and runs the sampling + per-image decode loop of measure_throughput on CPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_byte_compiles_and_runs_the_throughput_loop():
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import build_ref
    out = build_ref.build(verbose=False)
    if out is None and not os.path.isdir(os.path.join(ROOT, 'oracle', '_ref', 'rqvae')):
        pytest.skip('neither /root/reference nor a built oracle/_ref on this box')
    ref = os.path.join(ROOT, 'oracle', '_ref')
    assert not [f for _, _, fs in os.walk(ref) for f in fs if f.endswith('.py')], 'oracle/_ref must hold build outputs only, no sources'
    from rqvae import presets
    rqt = presets.rqtransformer_arch(128, 2, 2, 2, 500, vocab_size_cond=10, block_size=(8, 8, 4), input_embed_dim=64)
    arch = {'rqt': rqt, 'vae': {k: presets.RQVAE['tiny'][k] for k in ('hparams', 'ddconfig')}}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'ref_cpu_baseline.py'), '--arch', json.dumps(arch), '--batch', '3',
                        '--top-k', '50', '--top-p', '0.95'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['batch'] == 3 and d['pixels_shape'] == [3, 3, 16, 16] and d['codes_in_range'] and d['images_per_sec'] > 0
