"""RCCL (backend 'nccl' on ROCm) over xGMI: two ranks, one MI355X each, the data-parallel sampling path of
main_sampling_fid.py:166-169,196-227 on the real kernels -- per-rank seeds (seed + rank), every rank samples and decodes its
own batch with a tiny model pair, ONE all-gather of the decoded pixels (rqvae.utils.dist.all_gather_cat =
all_gather_into_tensor), rank-major order.  Skipped on boxes with fewer than two GPUs (the gpurun lease has one); the
CPU/gloo twin of this test is tests/test_dist_gloo.py.  Run with -m gpu."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'rq-vae-transformer_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    from types import SimpleNamespace
    import oracle
    from oracle import configs as C
    from rqvae.models.rqtransformer import RQTransformer
    from rqvae.models.rqvae import RQVAE
    from rqvae.utils import dist as dist_utils
    from rqvae.utils.utils import set_seed
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    distenv = dist_utils.initialize(SimpleNamespace(dist_backend='nccl', timeout=300))
    assert distenv.world_size == world and distenv.world_rank == rank
    hps, dd = C.VAE_TINY
    vae = RQVAE(**hps, ddconfig=dd, checkpointing=False)
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_params(oracle.rqvae_param_shapes(hps, dd), 31).items()})
    vae = vae.to(dev).eval()
    ar = RQTransformer(C.RQT_TINY)
    ar.load_state_dict({k: torch.from_numpy(v) for k, v in oracle.make_params(oracle.rqt_param_shapes(C.RQT_TINY), 41).items()})
    ar = dist_utils.dataparallel_and_sync(distenv, ar.to(dev).eval())          # DDP wrap + parameter broadcast (dist.py:70-85)
    B = 3
    cond = torch.tensor([[1], [2], [3]], device=dev) + 3 * rank

    def one_batch():
        set_seed(500 + distenv.world_rank)                                     # main_sampling_fid.py:166-169
        part = torch.zeros((B, 4, 4, 4), dtype=torch.long, device=dev)
        codes = ar.module.sample(part, vae, cond=cond, top_k=50, top_p=0.95)
        # the tiny transformer emits 4x4 code maps, the tiny VAE decodes 8x8: tile them (stand-in for matched shapes)
        codes8 = codes.repeat(1, 2, 2, 1).contiguous()
        pixels = torch.cat([vae.decode_code(codes8[i:i + 1]) for i in range(B)], dim=0)
        pixels = torch.clamp(pixels * 0.5 + 0.5, 0, 1)
        return codes, pixels, dist_utils.all_gather_cat(distenv, pixels), dist_utils.all_gather_cat(distenv, cond)
    codes, mine, gathered, targets = one_batch()
    codes_b, mine_b, gathered_b, _ = one_batch()
    assert torch.equal(codes, codes_b) and torch.equal(gathered, gathered_b)    # per-rank reproducible under seed + rank
    assert gathered.shape[0] == world * B and torch.equal(gathered[rank * B:(rank + 1) * B], mine)      # rank-major slots
    torch.distributed.barrier()
    if rank == 0:
        q.put(dict(targets=targets.flatten().tolist(), pix=gathered.cpu().numpy(), codes0=codes.cpu().numpy()))
    else:
        q.put(dict(rank=rank, mine=mine.cpu().numpy(), codes=codes.cpu().numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_rccl_sample_decode_gather():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL over xGMI); the single-GPU lease runs the gloo twin instead')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0 = next(g for g in got if 'pix' in g)
    r1 = next(g for g in got if 'mine' in g)
    assert r0['targets'] == [1, 2, 3, 4, 5, 6]                                   # rank-major label order
    assert np.array_equal(r0['pix'][3:], r1['mine'])                             # rank 1's pixels arrived bit-exact in slot 1
    assert not np.array_equal(r0['codes0'], r1['codes'])                         # seed + rank: different samples per rank
