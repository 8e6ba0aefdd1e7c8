"""rqvae/utils/dist.py:20-103 of the reference: env:// process group (backend 'nccl' == RCCL on ROCm),
DDP wrapping with explicit parameter broadcast, list all-gather + cat in rank order.

Sampling is embarrassingly parallel over images: every rank holds a full replica and
samples its own batch; the only data-path collective is the all-gather of decoded pixels
(main_sampling_fid.py:226), which goes over xGMI through RCCL."""
import datetime
import os
from dataclasses import dataclass

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


@dataclass
class DistEnv:
    world_size: int
    world_rank: int
    local_rank: int
    num_gpus: int
    master: bool
    device_name: str


def _device_name():
    return torch.cuda.get_device_name() if torch.cuda.is_available() else 'cpu'


def initialize(args, logger=None):
    """dist.py:30-67"""
    args.rank = int(os.environ.get("RANK", 0))
    args.world_size = int(os.environ.get('WORLD_SIZE', 1))
    args.local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if args.world_size > 1:
        os.environ["RANK"] = str(args.rank)
        os.environ["WORLD_SIZE"] = str(args.world_size)
        os.environ["LOCAL_RANK"] = str(args.local_rank)
        print(f'[dist] Distributed: wait dist process group:{args.local_rank}')
        dist.init_process_group(backend=getattr(args, 'dist_backend', 'nccl'), init_method='env://',
                                world_size=args.world_size,
                                timeout=datetime.timedelta(0, getattr(args, 'timeout', 86400)))
        assert args.world_size == dist.get_world_size()
        print(f"[dist] Distributed: success device:{args.local_rank}, {dist.get_rank()}/{dist.get_world_size()}")
        distenv = DistEnv(world_size=dist.get_world_size(), world_rank=dist.get_rank(), local_rank=args.local_rank,
                          num_gpus=1, master=(dist.get_rank() == 0), device_name=_device_name())
    else:
        print('[dist] Single processed')
        distenv = DistEnv(1, 0, 0, torch.cuda.device_count(), True, _device_name())
    print(f'[dist] {distenv}')
    if logger is not None:
        logger.info(distenv)
    return distenv


def dataparallel_and_sync(distenv, model, find_unused_parameters=False):
    """dist.py:70-85"""
    if dist.is_initialized():
        on_gpu = next(model.parameters()).is_cuda
        model = DistributedDataParallel(model, device_ids=[distenv.local_rank] if on_gpu else None,
                                        output_device=distenv.local_rank if on_gpu else None,
                                        find_unused_parameters=find_unused_parameters)
        for _, param in model.state_dict().items():
            dist.broadcast(param, 0)
        dist.barrier()
    else:
        # single process: the reference wraps in nn.DataParallel and only ever goes through `.module`
        # (main_sampling_fid.py:184,202,210).  Pinned to the model's own device: the native engine handle belongs to one
        # device and must never be shared by replicas on several.
        p = next(model.parameters(), None)
        ids = [p.device.index if p.device.index is not None else torch.cuda.current_device()] if p is not None and p.is_cuda else None
        model = torch.nn.DataParallel(model, device_ids=ids)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return model


def param_sync(param):
    dist.broadcast(param, 0)
    dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


@torch.no_grad()
def all_gather_cat(distenv, tensor, dim=0):
    """dist.py:94-103: rank-major concatenation.  One collective per call: the tensor form
    all_gather_into_tensor writes every rank's shard straight into its slot of the result (the
    reference gathers into a python list and then torch.cat's -- an extra full copy)."""
    if distenv.world_size == 1:
        return tensor
    t = tensor.contiguous()
    if dim == 0:
        out = torch.empty((distenv.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out
    g_tensor = [torch.empty_like(t) for _ in range(distenv.world_size)]
    dist.all_gather(g_tensor, t)
    return torch.cat(g_tensor, dim=dim)
