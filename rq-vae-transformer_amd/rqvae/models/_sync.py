"""Parameter -> engine synchronisation shared by RQVAE and RQTransformer.

The engines keep packed bf16 copies of the nn.Parameters.  A cheap signature (storage pointer and
in-place version counter of every tensor) decides when they must be pushed again: after
``load_state_dict``, ``.to(device)``, ``.half()/.float()`` or any in-place edit."""
import torch


def signature(module):
    sig = []
    for k, v in module.state_dict(keep_vars=True).items():
        sig.append((k, v.data_ptr(), v._version, v.device, v.dtype))
    return tuple(sig)


def push_all(module, engine, skip_prefixes=()):
    with torch.no_grad():
        for k, v in module.state_dict(keep_vars=True).items():
            if any(k.startswith(p) for p in skip_prefixes):
                continue
            engine.set_param(k, v)
