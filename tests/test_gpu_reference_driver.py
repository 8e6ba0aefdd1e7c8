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


def test_unchanged_main_sampling_fid_runs_its_sampling_loop_on_the_gpu(tmp_path):
    """The reference's UNCHANGED FID-sampling driver (`main_sampling_fid.py`, bytecode under oracle/_ref) on the MI355X: ITS load_model on
    checkpoint directories (config.yaml + {'state_dict': ...}; synthetic tiny models, no released checkpoint is reachable offline), ITS loop --
    `model_ar.module.sample(..., amp=True, fast=True)`, one `decode_code(pixels[i:i+1])` call per image, `all_gather_cat`, `save_pickle` of
    the [0, 1] pixels and the labels (:205-241).  Nothing is patched; tensorboard / torchvision / omegaconf come from tests/stubs.  The script's
    LAST step, compute_metrics (Inception / FID statistics: downloads), has nothing to work with offline; everything before it is checked:
    "[state] end of sampling", the pickles hold exactly what this package's API produces for the same seed, bit for bit, and the reference's
    own reader (rqvae/metrics/fid.py create_dataset_from_files) reads them back."""
    import glob
    import pickle

    import numpy as np
    import torch
    import yaml
    if not os.path.exists(os.path.join(REF, 'main_sampling_fid.pyc')):
        pytest.skip('oracle/_ref without main_sampling_fid (python oracle/build_ref.py, in the build container)')
    sys.path.insert(0, ROOT)
    import oracle
    from oracle import configs as C
    hps, dd = C.VAE_TINY
    d1, d2 = tmp_path / 'exp' / 'stage1', tmp_path / 'exp' / 'stage2'
    d1.mkdir(parents=True)
    d2.mkdir(parents=True)
    yaml.safe_dump({'arch': {'type': 'rq-vae', 'code_hier': 1, 'hparams': hps, 'ddconfig': dd}}, open(d1 / 'config.yaml', 'w'))
    vp = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), 31)
    torch.save({'state_dict': {k: torch.from_numpy(v) for k, v in vp.items()}}, d1 / 'model.pt')
    cfg = dict(C.RQT_TINY, block_size=list(hps['code_shape']))          # 8 x 8 x 4 codes: the tiny RQ-VAE's code shape
    yaml.safe_dump({'arch': cfg, 'sampling': {'temp': 1.0, 'top_k': 5, 'top_p': 0.9}, 'dataset': {'type': 'imagenet'}}, open(d2 / 'config.yaml', 'w'))
    ap = oracle.make_params(oracle.rqt_param_shapes(cfg), 41)
    torch.save({'state_dict': {k: torch.from_numpy(v) for k, v in ap.items()}}, d2 / 'epoch3_model.pt')

    env = dict(os.environ)
    env['PYTHONPATH'] = os.path.join(ROOT, 'tests', 'stubs')
    env['RQVAE_REFERENCE_ROOT'] = REF
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'rq-vae-transformer_amd', 'rqamd_run.py'), '-m', 'main_sampling_fid',
                        '-a', str(d2 / 'epoch3_model.pt'), '-v', str(d1 / 'model.pt'), '-n', '20', '-bs', '10', '--no-tensorboard',
                        '--no-stats-saving', '--save-dir', str(tmp_path / 'out'), '--seed', '7'],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    log = r.stdout + r.stderr
    assert '[state] end of sampling.' in log, log[-4000:]
    # (the script's last step, compute_metrics, finds no reference statistics for the synthetic run and returns without a metric:
    # exit code 0 on the MI355X box; should a future image make it raise, it must at least be the step that raises)
    assert r.returncode == 0 or 'compute_metrics' in log, log[-2000:]
    files = sorted(glob.glob(str(tmp_path / 'out' / '**' / 'samples_*.pkl'), recursive=True))
    assert [os.path.basename(f) for f in files] == ['samples_(1_2).pkl', 'samples_(2_2).pkl'], files
    got = [pickle.load(open(f, 'rb')) for f in files]
    tg = [np.load(f.replace('samples_', 'targets_').replace('.pkl', '.npz'))['targets'] for f in files]
    assert all(g.ndim == 4 and g.shape[:2] == (10, 3) and g.dtype == np.float32 and g.min() >= 0 and g.max() <= 1 for g in got)
    assert tg[0].tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4] and tg[1].tolist() == [5, 5, 6, 6, 7, 7, 8, 8, 9, 9]

    # the same through this package's API, same seed: what the driver's loop must have computed
    sys.path.insert(0, os.path.join(ROOT, 'rq-vae-transformer_amd'))
    from rqvae.models import create_model
    from rqvae.utils.config import load_config, augment_arch_defaults
    from rqvae.utils.utils import set_seed

    def load(path):
        config = load_config(os.path.join(os.path.dirname(path), 'config.yaml'))
        config.arch = augment_arch_defaults(config.arch)
        model, _ = create_model(config.arch, ema=False)
        model.load_state_dict(torch.load(path, map_location='cpu')['state_dict'])
        return model
    set_seed(7)
    ar, vae = load(str(d2 / 'epoch3_model.pt')).to('cuda').eval(), load(str(d1 / 'model.pt')).to('cuda').eval()
    conds = torch.arange(0, 10).repeat_interleave(2).reshape(2, 1, 10)
    for b in range(2):
        part = torch.zeros(10, *ar.get_block_size(), dtype=torch.long, device='cuda')
        codes = ar.sample(part, vae, cond=conds[b, 0].to('cuda'), temperature=1.0, top_k=5, top_p=0.9, amp=True, fast=True, is_tqdm=False)
        px = torch.cat([vae.decode_code(codes[i:i + 1]) for i in range(10)], 0)
        px = torch.clamp(px * 0.5 + 0.5, 0, 1)
        assert np.array_equal(px.cpu().numpy(), got[b]), b
    print('unchanged main_sampling_fid: 2 batches x 10 samples sampled, decoded one image per call, gathered and pickled by ITS loop == the API, bit for bit')

    # ... and the reference's reader takes the files the driver wrote
    code = ("import sys; sys.path[:0] = [%r]; import rqvae; from rqvae.metrics.fid import create_dataset_from_files as f; "
            "ds = f(%r); print('READ', len(ds), tuple(ds[0][0].shape))") % (os.path.join(ROOT, 'rq-vae-transformer_amd'), os.path.dirname(files[0]))
    r2 = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, timeout=300)
    assert f'READ 20 {tuple(got[0].shape[1:])}' in r2.stdout, r2.stdout[-1000:] + r2.stderr[-2000:]
