"""The reference's UNCHANGED drivers against this package.

CPU-only, build container only (needs /root/reference; skipped elsewhere): a subprocess puts this repo's `rqvae` mirror
first on sys.path, then tests/stubs (omegaconf / easydict / torchvision / clip stand-ins -- test infrastructure, see
tests/stubs/README.md), then the reference checkout, and

  * imports the reference's `measure_throughput/__main__.py` as a module and calls ITS `create_model` for every
    RQ-VAE (f32/f16/f8) x RQ-Transformer size it defines (meta device), checking the parameter counts the script prints;
  * imports the reference's `main_sampling_fid.py` and calls ITS `load_model` on a synthetic checkpoint directory
    (`config.yaml` + `{'state_dict': ...}`), for a stage-1 and a stage-2 model;
  * resolves `rqvae.metrics.fid` / `rqvae.img_datasets` to the reference's own files through the extended package path;
  * runs the launcher `rqamd_run.py -m measure_throughput` far enough to build the models (the timed loop needs a GPU)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('RQVAE_REFERENCE_ROOT', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'measure_throughput')), reason='reference checkout not present')


def run_py(code, *argv, cwd=None):
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'rq-vae-transformer_amd'), os.path.join(ROOT, 'tests', 'stubs'), REF])
    env['RQVAE_REFERENCE_ROOT'] = REF
    r = subprocess.run([sys.executable, '-c', textwrap.dedent(PRELUDE) + textwrap.dedent(code), *argv], capture_output=True, text=True, env=env, cwd=cwd or ROOT,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + '\n' + r.stderr[-3000:]
    return r.stdout


PRELUDE = '''
import sys, types, os, json
# torch.utils.tensorboard needs the tensorboard package at import time (main_sampling_fid.py:26)
tb = types.ModuleType('torch.utils.tensorboard'); tb.SummaryWriter = type('SummaryWriter', (), {'__init__': lambda s, *a, **k: None})
import torch
sys.modules['torch.utils.tensorboard'] = tb
import rqvae
assert 'rq-vae-transformer_amd' in rqvae.__file__, rqvae.__file__
'''


def test_measure_throughput_create_model_all_sizes():
    out = run_py('''
    import runpy
    ns = runpy.run_module('measure_throughput.__main__', run_name='measure_throughput_unchanged')
    import rqvae.models
    assert 'rq-vae-transformer_amd' in rqvae.models.__file__
    res = {}
    with torch.device('meta'):
        combos = [(f, hw, name, 4, 16384) for f, hw in (('f32', 8), ('f16', 16)) for name in ('small', 'medium', 'large', 'huge')]
        combos += [('f16', 16, 'vqgan_large', 1, 1024), ('f16', 16, 'vqgan_huge', 1, 16384)]     # depth-1 "VQ-GAN" shapes (:166-210)
        assert {c[2] for c in combos} == set(ns['RQTRANSFORMERS'])
        for f, hw, name, depth, K in combos:
            aux, ar = ns['create_model'](f, name, depth, K)
            assert list(aux.code_shape) == [hw, hw, depth]
            assert tuple(ar.block_size) == (hw, hw, depth) and ar.block_size_cond == 1
            res[f + '-' + name] = [sum(p.numel() for p in aux.parameters()) / 1e6, sum(p.numel() for p in ar.parameters()) / 1e6]
        aux, ar = ns['create_model']('f8', 'small', 2, 2048)
        res['f8-small-d2'] = [sum(p.numel() for p in aux.parameters()) / 1e6, sum(p.numel() for p in ar.parameters()) / 1e6]
    print('RESULT ' + json.dumps(res))
    ''')
    res = json.loads(out.split('RESULT ', 1)[1])
    # reference README.md:38-47 / the sizes the script is named after
    assert abs(res['f32-huge'][1] - 1387.5) < 0.1 and abs(res['f32-large'][1] - 820.9) < 0.1
    assert abs(res['f32-huge'][0] - 104.4 - 3 * 4.19) < 0.2 or abs(res['f32-huge'][0] - 104.4) < 0.2
    assert len(res) == 2 * 4 + 2 + 1


def test_main_sampling_fid_load_model(tmp_path):
    out = run_py('''
    import yaml
    sys.path.insert(0, %r)
    from oracle import configs as C
    import oracle
    tmp = sys.argv[1]
    import main_sampling_fid as drv                      # the reference's file, unchanged
    assert drv.__file__.startswith(%r)
    from rqvae.metrics.fid import compute_statistics_from_files          # resolves to the reference's module
    import rqvae.metrics.fid as fid
    assert fid.__file__.startswith(%r), fid.__file__
    hps, dd = C.VAE_TINY
    # stage 1: config.yaml as the reference's trainer writes it (arch under 'arch'), ckpt = {'state_dict': ...}
    d1 = os.path.join(tmp, 'stage1'); os.makedirs(d1)
    yaml.safe_dump({'arch': {'type': 'rq-vae', 'code_hier': 1, 'hparams': hps, 'ddconfig': dd}}, open(os.path.join(d1, 'config.yaml'), 'w'))
    vp = oracle.make_params(oracle.rqvae_param_shapes(hps, dd), 31)
    torch.save({'state_dict': {k: torch.from_numpy(v) for k, v in vp.items()}, 'epoch': 3}, os.path.join(d1, 'model.pt'))
    m1, cfg1 = drv.load_model(os.path.join(d1, 'model.pt'))
    assert cfg1.arch.checkpointing is False and cfg1.arch.hparams.use_padding_idx is False and cfg1.arch.ema is None
    assert type(m1).__module__.startswith('rqvae.models.rqvae') and 'rq-vae-transformer_amd' in sys.modules[type(m1).__module__].__file__
    sd = m1.state_dict()
    assert all(torch.equal(sd[k], torch.from_numpy(v)) for k, v in vp.items()) and set(sd) == set(vp)
    # stage 2 (+ EMA weights key)
    d2 = os.path.join(tmp, 'stage2'); os.makedirs(d2)
    cfg = C.RQT_TINY
    yaml.safe_dump({'arch': cfg}, open(os.path.join(d2, 'config.yaml'), 'w'))
    ap = oracle.make_params(oracle.rqt_param_shapes(cfg), 41)
    torch.save({'state_dict': {k: torch.from_numpy(v) for k, v in ap.items()},
                'state_dict_ema': {k: torch.from_numpy(v * 0 + 1) for k, v in ap.items()}}, os.path.join(d2, 'epoch3_model.pt'))
    m2, cfg2 = drv.load_model(os.path.join(d2, 'epoch3_model.pt'))
    assert cfg2.arch.vocab_size_cond == 10 and m2.get_block_size() == torch.Size([4, 4, 4])
    assert torch.equal(m2.state_dict()['pos_emb_hw'], torch.from_numpy(ap['pos_emb_hw']))
    m3, _ = drv.load_model(os.path.join(d2, 'epoch3_model.pt'), ema=True)
    assert float(m3.state_dict()['pos_emb_hw'].min()) == 1.0
    # what main() does next with the models (main_sampling_fid.py:184-202), minus the GPU: DataParallel wrap + .module
    import rqvae.utils.dist as dist_utils
    class A: pass
    distenv = dist_utils.initialize(A())
    wrapped = dist_utils.dataparallel_and_sync(distenv, m2)
    assert wrapped.module.get_block_size() == torch.Size([4, 4, 4])
    print('RESULT ok')
    ''' % (ROOT, REF, REF), str(tmp_path))
    assert 'RESULT ok' in out


def test_sample_files_round_trip_through_reference_reader(tmp_path):
    """Wire format either side of the path: what main_sampling_fid.py:231-241 writes per batch -- the
    gathered fp32 NCHW pixels in [0,1] pickled as `samples_(i_n).pkl` with rqvae.utils.utils.save_pickle (this repo's) and
    the labels as `targets_(i_n).npz` -- must be readable by the REFERENCE's own consumer, rqvae/metrics/fid.py
    create_dataset_from_files (which globs samples*.pkl and must not pick up the npz files)."""
    out = run_py('''
    import numpy as np
    from rqvae.utils.utils import save_pickle, set_seed                  # this repo's
    import rqvae.utils.utils as U
    assert 'rq-vae-transformer_amd' in U.__file__
    from rqvae.metrics.fid import create_dataset_from_files               # the reference's
    tmp = sys.argv[1]
    set_seed(3)
    n_batches, world, B = 3, 2, 4
    for i in range(n_batches):
        pixels = torch.rand(world * B, 3, 8, 8)                           # what all_gather_cat returns (rank-major), fp32 in [0,1]
        targets = torch.arange(world * B) % 5
        save_pickle(os.path.join(tmp, f'samples_({i+1}_{n_batches}).pkl'), pixels.cpu().numpy())
        np.savez(os.path.join(tmp, f'targets_({i+1}_{n_batches}).npz'), targets=targets.cpu().numpy())
    ds = create_dataset_from_files(tmp)
    assert len(ds) == n_batches * world * B
    x = ds[5][0]
    assert x.dtype == torch.float32 and tuple(x.shape) == (3, 8, 8) and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    t = np.load(os.path.join(tmp, 'targets_(1_3).npz'))['targets']
    assert t.dtype == np.int64 and t.shape == (8,)
    print('RESULT ok')
    ''', str(tmp_path))
    assert 'RESULT ok' in out


def test_launcher_runs_measure_throughput_to_the_gpu_boundary():
    """rqamd_run.py -m measure_throughput: import order, CLI parsing and model construction of the unchanged script; it must
    stop only where the script moves the models to 'cuda' (no GPU in this container)."""
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'tests', 'stubs')])
    env['RQVAE_REFERENCE_ROOT'] = REF
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'rq-vae-transformer_amd', 'rqamd_run.py'), '-m', 'measure_throughput',
                        'model=small', 'f=32', 'd=4', 'c=512', 'batch_size=2', 'n_loop=1', 'warmup=0'],
                       capture_output=True, text=True, env=env, cwd=REF, timeout=600)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr[-2000:]
        assert 'ms/sample' in r.stdout
    else:
        tail = r.stderr[-2500:]
        assert r.returncode != 0 and 'measure_throughput/__main__.py' in tail, tail
        assert 'model_aux.to(device)' in tail or 'cuda' in tail.lower(), tail
