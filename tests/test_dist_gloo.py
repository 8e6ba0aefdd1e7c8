"""CPU-only, 2 processes over gloo: the data-parallel sharding of main_sampling_fid.py (labels
reshaped (num_batches, world, B), rank takes [:, rank]; seeds = seed + rank; rank-major all-gather of
pixels and labels, main_sampling_fid.py:166-169,196-227) through rqvae.utils.dist."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'rq-vae-transformer_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from types import SimpleNamespace
    from rqvae.utils import dist as dist_utils
    from rqvae.utils.utils import set_seed
    args = SimpleNamespace(dist_backend='gloo', timeout=120)
    distenv = dist_utils.initialize(args)
    assert distenv.world_size == world and distenv.world_rank == rank and distenv.master == (rank == 0)
    seed = set_seed(100 + distenv.world_rank)
    seeds = dist_utils.all_gather_cat(distenv, torch.tensor([seed]))
    # label sharding exactly as the driver does it
    n_labels, n_samples, B = 4, 16, 2
    labels = torch.arange(n_labels).repeat_interleave(n_samples // n_labels)
    all_conds = labels.reshape(n_samples // (B * world), world, B)
    outs, tgts = [], []
    for batch_idx in range(all_conds.shape[0]):
        cond = all_conds[batch_idx, distenv.world_rank]
        pixels = cond.float().reshape(B, 1, 1, 1).expand(B, 3, 2, 2) + 0.01 * rank       # stand-in for decoded pixels
        outs.append(dist_utils.all_gather_cat(distenv, pixels.contiguous()))
        tgts.append(dist_utils.all_gather_cat(distenv, cond.contiguous()))
    # DDP wrap of a parameterised model + explicit broadcast (dist.py:70-85)
    lin = torch.nn.Linear(4, 4)
    with torch.no_grad():
        lin.weight.fill_(float(rank))
    wrapped = dist_utils.dataparallel_and_sync(distenv, lin)
    w_after = wrapped.module.weight.detach().clone()
    dist.barrier()
    if rank == 0:
        q.put(dict(seeds=seeds.tolist(), targets=torch.cat(tgts).tolist(), labels=labels.tolist(),
                   pix0=torch.cat(outs)[:, 0, 0, 0].tolist(), w=float(w_after.mean())))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res['seeds'] == [100, 101]
    assert res['targets'] == res['labels']                       # rank-major gather restores label order
    want = [float(lbl) + 0.01 * ((i // 2) % 2) for i, lbl in enumerate(res['labels'])]
    assert all(abs(a - b) < 1e-6 for a, b in zip(res['pix0'], want))
    assert res['w'] == 0.0                                        # rank 0's parameters were broadcast


def _run_bench(args, env_extra=None):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus2_self_launch_dry_run():
    """`python bench.py --gpus 2` with no torchrun environment must start two ranks itself (one process per GPU in the
    real run; gloo on CPU here, --dry-run = no kernels) and report the world size it actually joined."""
    out = _run_bench(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '4', '--dry-run'])
    assert out['n_gpus'] == 2 and out['world_size'] == 2 and out['requested_gpus'] == 2
    assert out['steps'] == 2 and out['warmup'] == 1 and out['scaling'] == 'weak' and out['gather_rank_major_ok'] is True
    assert out['config']['global_batch'] == 8
    one = _run_bench(['--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '4', '--dry-run'])
    assert one['n_gpus'] == 1
